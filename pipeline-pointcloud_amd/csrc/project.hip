// Fused per-Gaussian stage of the 3DGS hot path: activations -> quat/scale covariance ->
// EWA projection -> cull -> SH colour, forward and backward.  gfx950 only.
//
// Replaces (reference reaches these only through main.py:1312 / main.py:1343):
//   gsplat fully_fused_projection_{fwd,bwd}, quat_scale_to_covar_preci,
//   spherical_harmonics_{fwd,bwd}, and the torch.exp / torch.sigmoid / clamp(+0.5) glue
//   around them.
//
// Roofline: HBM streaming.  One thread per (camera, Gaussian) forward; one thread per
// Gaussian (looping cameras, no atomics) backward.  Algorithmic bytes: forward
// 44 B (+192 B SH when visible) read + 72 B written per visible Gaussian; backward
// 236 + 64 B read, 236 B written.
#include "common.h"
#include <math.h>
#include <stdlib.h>

namespace {

struct Cam {
    float R[9];
    float t[3];
    float fx, fy, cx, cy;
};

__device__ __forceinline__ Cam load_cam(const float* __restrict__ viewmats, const float* __restrict__ Ks, int c) {
    Cam cam;
    const float* V = viewmats + 16 * c;
    const float* K = Ks + 9 * c;
#pragma unroll
    for (int i = 0; i < 3; i++) {
#pragma unroll
        for (int j = 0; j < 3; j++) cam.R[3 * i + j] = V[4 * i + j];
        cam.t[i] = V[4 * i + 3];
    }
    cam.fx = K[0]; cam.fy = K[4]; cam.cx = K[2]; cam.cy = K[5];
    return cam;
}

__device__ __forceinline__ void quat_to_rotmat(const float q[4], float R[9], float& inv_norm) {
    float n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    inv_norm = rsqrtf(fmaxf(n2, 1e-24f));
    float w = q[0] * inv_norm, x = q[1] * inv_norm, y = q[2] * inv_norm, z = q[3] * inv_norm;
    R[0] = 1.f - 2.f * (y * y + z * z); R[1] = 2.f * (x * y - w * z); R[2] = 2.f * (x * z + w * y);
    R[3] = 2.f * (x * y + w * z); R[4] = 1.f - 2.f * (x * x + z * z); R[5] = 2.f * (y * z - w * x);
    R[6] = 2.f * (x * z - w * y); R[7] = 2.f * (y * z + w * x); R[8] = 1.f - 2.f * (x * x + y * y);
}

// Sigma = (R S)(R S)^T, symmetric 3x3 stored full.
__device__ __forceinline__ void covar_world(const float R[9], const float s[3], float S[9]) {
    float M[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) M[3 * i + j] = R[3 * i + j] * s[j];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            S[3 * i + j] = M[3 * i] * M[3 * j] + M[3 * i + 1] * M[3 * j + 1] + M[3 * i + 2] * M[3 * j + 2];
}

// A(3x3) * B(3x3)
__device__ __forceinline__ void mat3_mul(const float A[9], const float B[9], float Cm[9]) {
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            Cm[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
// A * B^T
__device__ __forceinline__ void mat3_mul_bt(const float A[9], const float B[9], float Cm[9]) {
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            Cm[3 * i + j] = A[3 * i] * B[3 * j] + A[3 * i + 1] * B[3 * j + 1] + A[3 * i + 2] * B[3 * j + 2];
}
// A^T * B
__device__ __forceinline__ void mat3_mul_at(const float A[9], const float B[9], float Cm[9]) {
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            Cm[3 * i + j] = A[i] * B[j] + A[3 + i] * B[3 + j] + A[6 + i] * B[6 + j];
}

struct Proj {
    float mc[3];        // camera-space mean
    float Sc[9];        // camera-space covariance
    float J[6];         // 2x3 Jacobian
    float tx, ty;
    bool x_in, y_in;
    float a, b, c;      // blurred 2-D covariance
    float det, det_orig, comp;
    float conA, conB, conC;
    float m2x, m2y;
};

__device__ __forceinline__ void project_core(const Cam& cam, const float mean[3], const float Sw[9],
                                             int W, int H, float eps2d, Proj& P) {
#pragma unroll
    for (int i = 0; i < 3; i++)
        P.mc[i] = cam.R[3 * i] * mean[0] + cam.R[3 * i + 1] * mean[1] + cam.R[3 * i + 2] * mean[2] + cam.t[i];
    float tmp[9];
    mat3_mul(cam.R, Sw, tmp);
    mat3_mul_bt(tmp, cam.R, P.Sc);
    float x = P.mc[0], y = P.mc[1], z = P.mc[2];
    float rz = 1.f / z, rz2 = rz * rz;
    float tan_fovx = 0.5f * W / cam.fx, tan_fovy = 0.5f * H / cam.fy;
    float lim_x_pos = (W - cam.cx) / cam.fx + 0.3f * tan_fovx;
    float lim_x_neg = cam.cx / cam.fx + 0.3f * tan_fovx;
    float lim_y_pos = (H - cam.cy) / cam.fy + 0.3f * tan_fovy;
    float lim_y_neg = cam.cy / cam.fy + 0.3f * tan_fovy;
    float xz = x * rz, yz = y * rz;
    P.x_in = (xz <= lim_x_pos) && (xz >= -lim_x_neg);
    P.y_in = (yz <= lim_y_pos) && (yz >= -lim_y_neg);
    P.tx = z * fminf(lim_x_pos, fmaxf(-lim_x_neg, xz));
    P.ty = z * fminf(lim_y_pos, fmaxf(-lim_y_neg, yz));
    P.J[0] = cam.fx * rz; P.J[1] = 0.f; P.J[2] = -cam.fx * P.tx * rz2;
    P.J[3] = 0.f; P.J[4] = cam.fy * rz; P.J[5] = -cam.fy * P.ty * rz2;
    // cov2d = J Sc J^T
    float JS[6];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            JS[3 * i + j] = P.J[3 * i] * P.Sc[j] + P.J[3 * i + 1] * P.Sc[3 + j] + P.J[3 * i + 2] * P.Sc[6 + j];
    float a0 = JS[0] * P.J[0] + JS[1] * P.J[1] + JS[2] * P.J[2];
    float b0 = JS[0] * P.J[3] + JS[1] * P.J[4] + JS[2] * P.J[5];
    float c0 = JS[3] * P.J[3] + JS[4] * P.J[4] + JS[5] * P.J[5];
    P.det_orig = a0 * c0 - b0 * b0;
    P.a = a0 + eps2d; P.b = b0; P.c = c0 + eps2d;
    P.det = P.a * P.c - P.b * P.b;
    P.comp = sqrtf(fmaxf(0.f, P.det_orig / P.det));
    float rdet = 1.f / P.det;
    P.conA = P.c * rdet; P.conB = -P.b * rdet; P.conC = P.a * rdet;
    P.m2x = cam.fx * x * rz + cam.cx;
    P.m2y = cam.fy * y * rz + cam.cy;
}

constexpr float SH_C0 = 0.2820947917738781f;
constexpr float SH_C1 = 0.48860251190292f;
constexpr float SH_C2_0 = 1.0925484305920792f, SH_C2_1 = -1.0925484305920792f, SH_C2_2 = 0.31539156525252005f,
                SH_C2_3 = -1.0925484305920792f, SH_C2_4 = 0.5462742152960396f;
constexpr float SH_C3_0 = -0.5900435899266435f, SH_C3_1 = 2.890611442640554f, SH_C3_2 = -0.4570457994644658f,
                SH_C3_3 = 0.3731763325901154f, SH_C3_4 = -0.4570457994644658f, SH_C3_5 = 1.445305721320277f,
                SH_C3_6 = -0.5900435899266435f;

// basis b[0..nb) for unit direction (x,y,z)
__device__ __forceinline__ void sh_basis(int deg, float x, float y, float z, float b[16]) {
    b[0] = SH_C0;
    if (deg >= 1) { b[1] = -SH_C1 * y; b[2] = SH_C1 * z; b[3] = -SH_C1 * x; }
    if (deg >= 2) {
        float xx = x * x, yy = y * y, zz = z * z;
        b[4] = SH_C2_0 * x * y; b[5] = SH_C2_1 * y * z; b[6] = SH_C2_2 * (2.f * zz - xx - yy);
        b[7] = SH_C2_3 * x * z; b[8] = SH_C2_4 * (xx - yy);
        if (deg >= 3) {
            b[9] = SH_C3_0 * y * (3.f * xx - yy); b[10] = SH_C3_1 * x * y * z;
            b[11] = SH_C3_2 * y * (4.f * zz - xx - yy); b[12] = SH_C3_3 * z * (2.f * zz - 3.f * xx - 3.f * yy);
            b[13] = SH_C3_4 * x * (4.f * zz - xx - yy); b[14] = SH_C3_5 * z * (xx - yy);
            b[15] = SH_C3_6 * x * (xx - 3.f * yy);
        }
    }
}

// d(basis)/d(x,y,z) contracted with per-basis weights g[k] (= sum_ch v_rgb[ch]*coef[k][ch])
__device__ __forceinline__ void sh_basis_vjp(int deg, float x, float y, float z, const float g[16], float vd[3]) {
    float vx = 0.f, vy = 0.f, vz = 0.f;
    if (deg >= 1) { vy += -SH_C1 * g[1]; vz += SH_C1 * g[2]; vx += -SH_C1 * g[3]; }
    if (deg >= 2) {
        vx += SH_C2_0 * y * g[4]; vy += SH_C2_0 * x * g[4];
        vy += SH_C2_1 * z * g[5]; vz += SH_C2_1 * y * g[5];
        vx += SH_C2_2 * -2.f * x * g[6]; vy += SH_C2_2 * -2.f * y * g[6]; vz += SH_C2_2 * 4.f * z * g[6];
        vx += SH_C2_3 * z * g[7]; vz += SH_C2_3 * x * g[7];
        vx += SH_C2_4 * 2.f * x * g[8]; vy += SH_C2_4 * -2.f * y * g[8];
        if (deg >= 3) {
            float xx = x * x, yy = y * y, zz = z * z;
            // b9 = C*y*(3xx-yy)
            vx += SH_C3_0 * 6.f * x * y * g[9]; vy += SH_C3_0 * (3.f * xx - 3.f * yy) * g[9];
            // b10 = C*xyz
            vx += SH_C3_1 * y * z * g[10]; vy += SH_C3_1 * x * z * g[10]; vz += SH_C3_1 * x * y * g[10];
            // b11 = C*y*(4zz-xx-yy)
            vx += SH_C3_2 * -2.f * x * y * g[11]; vy += SH_C3_2 * (4.f * zz - xx - 3.f * yy) * g[11];
            vz += SH_C3_2 * 8.f * y * z * g[11];
            // b12 = C*z*(2zz-3xx-3yy)
            vx += SH_C3_3 * -6.f * x * z * g[12]; vy += SH_C3_3 * -6.f * y * z * g[12];
            vz += SH_C3_3 * (6.f * zz - 3.f * xx - 3.f * yy) * g[12];
            // b13 = C*x*(4zz-xx-yy)
            vx += SH_C3_4 * (4.f * zz - 3.f * xx - yy) * g[13]; vy += SH_C3_4 * -2.f * x * y * g[13];
            vz += SH_C3_4 * 8.f * x * z * g[13];
            // b14 = C*z*(xx-yy)
            vx += SH_C3_5 * 2.f * x * z * g[14]; vy += SH_C3_5 * -2.f * y * z * g[14]; vz += SH_C3_5 * (xx - yy) * g[14];
            // b15 = C*x*(xx-3yy)
            vx += SH_C3_6 * (3.f * xx - 3.f * yy) * g[15]; vy += SH_C3_6 * -6.f * x * y * g[15];
        }
    }
    vd[0] = vx; vd[1] = vy; vd[2] = vz;
}

__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + __expf(-x)); }

constexpr int SH_WAVE_F4 = 64 * 45 / 4;   // one wave's shN slice in float4 units (11 520 B)
constexpr int SH_HALF_F4 = 32 * 45 / 4;   // 32 rows of it (5 760 B)

// colour += sum over the bases 1 .. nb - 1 of b[k] * coefficient row k (45 floats in LDS).  All 45 LDS reads go out together:
// a loop over the runtime degree waits out one LDS round trip per coefficient.
__device__ __forceinline__ void sh_accumulate(const float* cN, const float (&b)[16], int nb, float (&rgb)[3]) {
    // five bases (15 floats) per LDS round trip: one at a time a loop over the runtime degree waits out 15 round trips, all 45
    // at once cost 45 live registers on top of the slice half still waiting in registers
#pragma unroll
    for (int g = 0; g < 3; g++) {
        float cf[15];
#pragma unroll
        for (int k = 0; k < 15; k++) cf[k] = cN[15 * g + k];
#pragma unroll
        for (int kk = 0; kk < 5; kk++) {
            const int k = 1 + 5 * g + kk;
            const float bk = k < nb ? b[k] : 0.f;
            rgb[0] += bk * (k < nb ? cf[3 * kk] : 0.f);
            rgb[1] += bk * (k < nb ? cf[3 * kk + 1] : 0.f);
            rgb[2] += bk * (k < nb ? cf[3 * kk + 2] : 0.f);
        }
    }
}

// --------------------------------------------------------------------------- forward
// color_mode: 0 = SH (sh0[N,3] + shN[N,15,3]), 1 = colors[N,3], 2 = colors[C,N,3]
// (waves_per_eu: with 23 KB of LDS the compiler otherwise aims at six waves per SIMD and gets there by parking the staged
// slice in scratch, one waited load at a time)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void project_fwd_kernel(
    int C, int N, const float* __restrict__ means, const float* __restrict__ quats,
    const float* __restrict__ scales, const float* __restrict__ opacities,
    const float* __restrict__ sh0, const float* __restrict__ shN, const float* __restrict__ colors,
    int color_mode, int sh_degree, const float* __restrict__ viewmats, const float* __restrict__ Ks,
    int W, int H, float eps2d, float near_plane, float far_plane, float radius_clip, int flags,
    int32_t* __restrict__ radii, float* __restrict__ splats, uint32_t* __restrict__ depth_keys) {
    // Each wave's 64 Gaussians own one contiguous 11 520-byte slice of shN.  Reading it as
    // 45 dword loads per lane at a 180-byte stride made the first version latency/TA bound
    // (62 VMEM instructions per wave); instead the wave copies the slice with 16-byte loads
    // into LDS and every lane reads its 45 coefficients from there (stride 45 dwords: odd,
    // bank-conflict free).
    // The slice goes through LDS in two halves of 32 rows (5 760 B per wave instead of 11 520: four waves per SIMD instead
    // of three); the second half's loads are in flight while the lanes of the first evaluate.  Measured 150 -> 145 us on one
    // box: the kernel is bound by its ~2 000 vector instructions per wave more than by latency (padding the LDS back to three
    // blocks per CU costs 2 us), and fetching only the rows of visible Gaussians is worth more than the index arithmetic it
    // costs (146 against 155 us without).
    __shared__ float4 sSH4[4][SH_HALF_F4];
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    const bool live = idx < (long long)C * N;
    long long idx_c = live ? idx : (long long)C * N - 1;
    int c = (int)(idx_c / N);
    int n = (int)(idx_c - (long long)c * N);
    float4* rec = reinterpret_cast<float4*>(splats + idx_c * SPLAT_STRIDE);
    int2* rad = reinterpret_cast<int2*>(radii + idx_c * 2);

    // one camera: a compile-time index, so the matrices arrive through scalar loads instead of a
    // per-lane gather that everything else would have to wait for
    Cam cam = (C == 1) ? load_cam(viewmats, Ks, 0) : load_cam(viewmats, Ks, c);
    float mean[3] = {means[3 * n], means[3 * n + 1], means[3 * n + 2]};
    // the opacity travels with the means (it used to be a separate round trip after the projection) -- and, since round 3, so do
    // the quaternion, the scales and the DC colour: they used to be requested behind the depth test (`if (ok)`), one more
    // dependent memory round trip per wave for the sake of 40 bytes of the culled third of the Gaussians (+27 MB of 441)
    const float opa_raw = opacities != nullptr ? opacities[n] : 1.f;
    const float4 q4 = *reinterpret_cast<const float4*>(quats + 4 * (long long)n);
    const float s_raw[3] = {scales[3 * n], scales[3 * n + 1], scales[3 * n + 2]};
    float c0[3] = {0.f, 0.f, 0.f};
    if (color_mode == 0) { c0[0] = sh0[3 * (long long)n]; c0[1] = sh0[3 * (long long)n + 1]; c0[2] = sh0[3 * (long long)n + 2]; }
    asm volatile("" :: "v"(q4.x), "v"(s_raw[0]), "v"(c0[0]));          // (keeps these loads in front of the branch below: hipcc sinks them into it)
    float zc = cam.R[6] * mean[0] + cam.R[7] * mean[1] + cam.R[8] * mean[2] + cam.t[2];
    bool ok = live && (zc >= near_plane) && (zc <= far_plane);
    Proj P;
    float opa = 1.f;
    float rx = 0.f, ry = 0.f;
    if (ok) {
        float q[4] = {q4.x, q4.y, q4.z, q4.w};
        float s[3] = {s_raw[0], s_raw[1], s_raw[2]};
        if (flags & MI_FLAG_LOG_SCALES) { s[0] = __expf(s[0]); s[1] = __expf(s[1]); s[2] = __expf(s[2]); }
        float Rq[9], inv_norm, Sw[9];
        quat_to_rotmat(q, Rq, inv_norm);
        covar_world(Rq, s, Sw);
        project_core(cam, mean, Sw, W, H, eps2d, P);
        ok = P.det > 0.f;
        float extend = 3.33f;
        if (ok && opacities != nullptr) {
            opa = opa_raw;
            if (flags & MI_FLAG_LOGIT_OPAC) opa = sigmoidf(opa);
            if (flags & MI_FLAG_ANTIALIASED) opa *= P.comp;
            if (opa < ALPHA_THRESHOLD) ok = false;
            else extend = fminf(extend, sqrtf(2.0f * __logf(opa / ALPHA_THRESHOLD)));
        }
        if (ok) {
            float bm = 0.5f * (P.a + P.c);
            float v1 = bm + sqrtf(fmaxf(0.01f, bm * bm - P.det));
            float r1 = extend * sqrtf(v1);
            rx = ceilf(fminf(extend * sqrtf(P.a), r1));
            ry = ceilf(fminf(extend * sqrtf(P.c), r1));
            if (rx <= radius_clip && ry <= radius_clip) ok = false;
            if (P.m2x + rx <= 0.f || P.m2x - rx >= (float)W || P.m2y + ry <= 0.f || P.m2y - ry >= (float)H) ok = false;
        }
    }
    // ---- stage this wave's shN slice (wave-uniform decision); the DC coefficients are requested in the
    // same breath (they used to be one more round trip after the slice)
    bool staged = false;
    float staged_rgb[3] = {0.f, 0.f, 0.f};          // the shN part of the colour when the slice went through LDS
    float b[16];                                    // SH basis of the view direction (zeros for a culled lane)
#pragma unroll
    for (int k = 0; k < 16; k++) b[k] = 0.f;
    if (color_mode == 0 && ok) {
        // campos = -R^T t ; dir = mean - campos
        float cp[3];
#pragma unroll
        for (int i = 0; i < 3; i++) cp[i] = -(cam.R[i] * cam.t[0] + cam.R[3 + i] * cam.t[1] + cam.R[6 + i] * cam.t[2]);
        float dx = mean[0] - cp[0], dy = mean[1] - cp[1], dz = mean[2] - cp[2];
        float inv = rsqrtf(fmaxf(dx * dx + dy * dy + dz * dz, 1e-24f));
        sh_basis(sh_degree, dx * inv, dy * inv, dz * inv, b);
    }
    if (color_mode == 0 && sh_degree >= 2) {
        long long idx0 = idx - lane;                                  // first record of the wave
        int n0 = (int)(idx0 - (long long)(idx0 / N) * N);
        bool whole = idx0 + 63 < (long long)C * N && (idx0 / N) == ((idx0 + 63) / N);   // one camera, 64 live lanes
        const float* src = shN + 45 * (long long)n0;
        const unsigned long long okm = wave_ballot(ok);
        if (whole && (((uintptr_t)src) & 15) == 0 && okm != 0ull) {
            staged = true;
            const float4* s4 = reinterpret_cast<const float4*>(src);
            // loads first (index clamped, no branch), LDS stores after: with the bounds test around each
            // copy the compiler serialised them, load -> s_waitcnt vmcnt(0) -> ds_write twelve times.
            // Only the rows of Gaussians that survived the culling are fetched (a third of them did not in the 2 M scene:
            // 120 MB of 625): a float4 none of whose rows is needed reads the slice's first 16 bytes instead (one address for
            // all such lanes, no branch).
            constexpr int HJ = (SH_HALF_F4 + 63) / 64;               // 6 loads per lane and half
            // (two plain arrays and straight-line code: as tmp[2][HJ] under a loop over the halves the compiler put the
            // values in scratch and waited for every load on its own)
            const int nb = (sh_degree + 1) * (sh_degree + 1);
            static_assert(HJ == 6, "six named registers per half below");
            // (named values, not an array: an array that is live across the wavefront fences stays a stack object -- the
            // compiler stored all of it to scratch at every fence)
            float4 u0, u1, u2, u3, u4, u5;
            auto fetch = [&](int j, unsigned okw, int ofs) -> float4 {
                const int i4 = min(lane + 64 * j, SH_HALF_F4 - 1);
                const int r0 = (4 * i4) / 45, r1 = (4 * i4 + 3) / 45;
                const bool need = ((okw >> r0) | (okw >> r1)) & 1u;
                return s4[need ? ofs + i4 : 0];
            };
            auto put = [&](int j, const float4& v) {
                const int i4 = lane + 64 * j;
                if (i4 < SH_HALF_F4) sSH4[wv][i4] = v;
            };
            const unsigned okl = (unsigned)okm, okh = (unsigned)(okm >> 32);
            u0 = fetch(0, okl, 0); u1 = fetch(1, okl, 0); u2 = fetch(2, okl, 0);
            u3 = fetch(3, okl, 0); u4 = fetch(4, okl, 0); u5 = fetch(5, okl, 0);
            put(0, u0); put(1, u1); put(2, u2); put(3, u3); put(4, u4); put(5, u5);
            // the second half is requested before the first is evaluated (the wavefront-scope fences order the LDS traffic of
            // this wave without waiting for the loads in flight)
            u0 = fetch(0, okh, SH_HALF_F4); u1 = fetch(1, okh, SH_HALF_F4); u2 = fetch(2, okh, SH_HALF_F4);
            u3 = fetch(3, okh, SH_HALF_F4); u4 = fetch(4, okh, SH_HALF_F4); u5 = fetch(5, okh, SH_HALF_F4);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");            // same-wave LDS hand-off: order only
            __builtin_amdgcn_wave_barrier();
            if (ok && lane < 32) sh_accumulate(reinterpret_cast<const float*>(sSH4[wv]) + 45 * lane, b, nb, staged_rgb);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");            // the first half has been read
            __builtin_amdgcn_wave_barrier();
            put(0, u0); put(1, u1); put(2, u2); put(3, u3); put(4, u4); put(5, u5);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (ok && lane >= 32) sh_accumulate(reinterpret_cast<const float*>(sSH4[wv]) + 45 * (lane - 32), b, nb, staged_rgb);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");            // same-wave LDS hand-off: order only
    if (!live) return;
    if (!ok) {
        *rad = make_int2(0, 0);
        float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        rec[0] = z4; rec[1] = z4; rec[2] = z4; rec[3] = z4;
        if (depth_keys) depth_keys[idx] = 0xFFFFFFFFu;      // culled: sorts last
        return;
    }
    float rgb[3];
    if (color_mode == 0) {
        rgb[0] = b[0] * c0[0]; rgb[1] = b[0] * c0[1]; rgb[2] = b[0] * c0[2];
        int nb = (sh_degree + 1) * (sh_degree + 1);
        if (staged) {
            rgb[0] += staged_rgb[0]; rgb[1] += staged_rgb[1]; rgb[2] += staged_rgb[2];
        } else {
            const float* cN = shN + 45 * (long long)n;
            for (int k = 1; k < nb; k++) {
                rgb[0] += b[k] * cN[3 * (k - 1)];
                rgb[1] += b[k] * cN[3 * (k - 1) + 1];
                rgb[2] += b[k] * cN[3 * (k - 1) + 2];
            }
        }
        rgb[0] = fmaxf(rgb[0] + 0.5f, 0.f); rgb[1] = fmaxf(rgb[1] + 0.5f, 0.f); rgb[2] = fmaxf(rgb[2] + 0.5f, 0.f);
    } else {
        const float* cc = colors + 3 * (color_mode == 2 ? idx : (long long)n);
        rgb[0] = cc[0]; rgb[1] = cc[1]; rgb[2] = cc[2];
    }
    *rad = make_int2((int)rx, (int)ry);
    rec[0] = make_float4(P.m2x, P.m2y, P.conA, P.conB);
    rec[1] = make_float4(P.conC, opa, rgb[0], rgb[1]);
    rec[2] = make_float4(rgb[2], P.mc[2], P.comp, __int_as_float((int)rx));
    rec[3] = make_float4(__int_as_float((int)ry), 0.f, 0.f, 0.f);
    if (depth_keys) depth_keys[idx] = __float_as_uint(P.mc[2]);   // depth > 0: the bit pattern orders like the value
}

// -------------------------------------------------------------------------- backward
// Shared per-(camera, Gaussian) backward of the geometry: everything except the colour path.
struct GeoGrad {
    float vmean[3];
    float vSw[9];
    float vopa;
};

// rec_conic: the conic the FORWARD stored in the splat record (what the rasteriser used, and what v_conic is the gradient
// of).  Recomputing it here is not just redundant: det = a c - b^2 cancels catastrophically for a needle-thin Gaussian that
// projects thousands of pixels wide (a ~ c ~ 1e7: the difference is all rounding), the forward's instruction sequence gave a
// small positive det and this kernel's, contracted differently, gave 0 -> 1 / det = inf -> 0 * inf = NaN in every
// geometric gradient even with an all-zero gradient record.  Found in the MCMC synthetic run (one Gaussian in ~5 000 steps; a
// NaN Gaussian is never visible again but keeps its opacity, is picked as a relocation source and spreads:
// tools/nan_probe.py reproduces the single call).
__device__ __forceinline__ void geo_bwd_one_camera(const Cam& cam, const float mean[3], const float Sw[9], int W, int H,
                                                   float eps2d, int flags, float opa_act, float v_m2x, float v_m2y,
                                                   float v_cA, float v_cB, float v_cC, float v_op, float v_depth,
                                                   float recA, float recB, float recC, GeoGrad& G) {
    Proj P;
    project_core(cam, mean, Sw, W, H, eps2d, P);
    P.conA = recA; P.conB = recB; P.conC = recC;
    float vcov_a = 0.f, vcov_b = 0.f, vcov_c = 0.f;   // symmetric 2x2 grad: [[a, b],[b, c]]
    if (flags & MI_FLAG_ANTIALIASED) {
        G.vopa += v_op * P.comp;
        float v_comp = v_op * opa_act;
        float det_conic = P.conA * P.conC - P.conB * P.conB;
        float v_sqr = v_comp * 0.5f / (P.comp + 1e-6f);
        float om = 1.f - P.comp * P.comp;
        vcov_a += v_sqr * (om * P.conA - eps2d * det_conic);
        vcov_b += v_sqr * (om * P.conB);
        vcov_c += v_sqr * (om * P.conC - eps2d * det_conic);
    } else {
        G.vopa += v_op;
    }
    // conic = inverse(cov2d): G_cov = -X G_X X, G_X = [[vA, vB/2],[vB/2, vC]]
    {
        float xa = P.conA, xb = P.conB, xc = P.conC;
        float ga = v_cA, gb = 0.5f * v_cB, gc = v_cC;
        float t00 = xa * ga + xb * gb, t01 = xa * gb + xb * gc;
        float t10 = xb * ga + xc * gb, t11 = xb * gb + xc * gc;
        vcov_a -= t00 * xa + t01 * xb;
        vcov_b -= t00 * xb + t01 * xc;
        vcov_c -= t10 * xb + t11 * xc;
    }
    // cov2d = J Sc J^T ; mean2d
    float Gm[4] = {vcov_a, vcov_b, vcov_b, vcov_c};
    float GJ[6], vSc[9], vJ[6];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) GJ[3 * i + j] = Gm[2 * i] * P.J[j] + Gm[2 * i + 1] * P.J[3 + j];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) vSc[3 * i + j] = P.J[i] * GJ[j] + P.J[3 + i] * GJ[3 + j];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            vJ[3 * i + j] = 2.f * (GJ[3 * i] * P.Sc[j] + GJ[3 * i + 1] * P.Sc[3 + j] + GJ[3 * i + 2] * P.Sc[6 + j]);
    float x = P.mc[0], y = P.mc[1], z = P.mc[2];
    float rz = 1.f / z, rz2 = rz * rz, rz3 = rz2 * rz;
    float vmc[3];
    vmc[0] = cam.fx * rz * v_m2x;
    vmc[1] = cam.fy * rz * v_m2y;
    vmc[2] = -(cam.fx * x * v_m2x + cam.fy * y * v_m2y) * rz2 + v_depth;
    if (P.x_in) vmc[0] += -cam.fx * rz2 * vJ[2];
    else vmc[2] += -cam.fx * rz3 * vJ[2] * P.tx;
    if (P.y_in) vmc[1] += -cam.fy * rz2 * vJ[5];
    else vmc[2] += -cam.fy * rz3 * vJ[5] * P.ty;
    vmc[2] += -cam.fx * rz2 * vJ[0] - cam.fy * rz2 * vJ[4] + 2.f * cam.fx * P.tx * rz3 * vJ[2] +
              2.f * cam.fy * P.ty * rz3 * vJ[5];
#pragma unroll
    for (int i = 0; i < 3; i++) G.vmean[i] += cam.R[i] * vmc[0] + cam.R[3 + i] * vmc[1] + cam.R[6 + i] * vmc[2];
    float tmp[9], acc[9];
    mat3_mul_at(cam.R, vSc, tmp);
    mat3_mul(tmp, cam.R, acc);
#pragma unroll
    for (int i = 0; i < 9; i++) G.vSw[i] += acc[i];
}

// last line of defence for the optimiser: a non-finite geometric gradient (overflow in a degenerate projection) is dropped
// for that Gaussian and step instead of being written into its parameters and moments for good
__device__ __forceinline__ void drop_nonfinite_geo(float (&vmean)[3], float (&vq)[4], float (&vs)[3]) {
    const float t = vmean[0] + vmean[1] + vmean[2] + vq[0] + vq[1] + vq[2] + vq[3] + vs[0] + vs[1] + vs[2];
    if (!__builtin_isfinite(t)) {
        vmean[0] = vmean[1] = vmean[2] = 0.f;
        vq[0] = vq[1] = vq[2] = vq[3] = 0.f;
        vs[0] = vs[1] = vs[2] = 0.f;
    }
}

// Sigma = M M^T, M = R(q) diag(s)  ->  v_quat (through the normalisation), v_scale
__device__ __forceinline__ void covar_bwd(const float Rq[9], const float s[3], const float q[4], float inv_norm,
                                          const float vSw[9], float vq[4], float vs[3]) {
    float M[9], vM[9], vR[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) M[3 * i + j] = Rq[3 * i + j] * s[j];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            vM[3 * i + j] = (vSw[3 * i] + vSw[i]) * M[j] + (vSw[3 * i + 1] + vSw[3 + i]) * M[3 + j] +
                            (vSw[3 * i + 2] + vSw[6 + i]) * M[6 + j];
#pragma unroll
    for (int j = 0; j < 3; j++) vs[j] = Rq[j] * vM[j] + Rq[3 + j] * vM[3 + j] + Rq[6 + j] * vM[6 + j];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) vR[3 * i + j] = vM[3 * i + j] * s[j];
    float w = q[0] * inv_norm, x = q[1] * inv_norm, y = q[2] * inv_norm, z = q[3] * inv_norm;
    float vqn[4];
    vqn[0] = 2.f * (x * (vR[7] - vR[5]) + y * (vR[2] - vR[6]) + z * (vR[3] - vR[1]));
    vqn[1] = 2.f * (-2.f * x * (vR[4] + vR[8]) + y * (vR[1] + vR[3]) + z * (vR[2] + vR[6]) + w * (vR[7] - vR[5]));
    vqn[2] = 2.f * (x * (vR[1] + vR[3]) - 2.f * y * (vR[0] + vR[8]) + z * (vR[5] + vR[7]) + w * (vR[2] - vR[6]));
    vqn[3] = 2.f * (x * (vR[2] + vR[6]) + y * (vR[5] + vR[7]) - 2.f * z * (vR[0] + vR[4]) + w * (vR[3] - vR[1]));
    float dq = vqn[0] * w + vqn[1] * x + vqn[2] * y + vqn[3] * z;
    vq[0] = (vqn[0] - dq * w) * inv_norm; vq[1] = (vqn[1] - dq * x) * inv_norm;
    vq[2] = (vqn[2] - dq * y) * inv_norm; vq[3] = (vqn[3] - dq * z) * inv_norm;
}

// copies a wave's slice of `count` floats between memory and LDS with 16-byte accesses (+ scalar tail)
__device__ __forceinline__ void slice_load(float* lds, const float* src, int count, int lane) {
    int n4 = count >> 2;
    const float4* s4 = reinterpret_cast<const float4*>(src);
    float4* l4 = reinterpret_cast<float4*>(lds);
    float4 tmp[(SH_WAVE_F4 + 63) / 64];
#pragma unroll
    for (int j = 0; j < (SH_WAVE_F4 + 63) / 64; j++) tmp[j] = s4[min(lane + 64 * j, max(n4 - 1, 0))];   // all loads in flight
    // (the clobber keeps the twelve loads in FRONT of the LDS stores: without it hipcc pairs each load with its store again,
    //  load -> s_waitcnt vmcnt(0) -> ds_write twelve times in a row, in the unfused backward the MCMC and data-parallel trainers run)
    asm volatile("" ::: "memory");
#pragma unroll
    for (int j = 0; j < (SH_WAVE_F4 + 63) / 64; j++) {
        int i4 = lane + 64 * j;
        if (i4 < n4) l4[i4] = tmp[j];
    }
    int rem = count & 3;
    if (lane < rem) lds[4 * n4 + lane] = src[4 * n4 + lane];
}
__device__ __forceinline__ void slice_store(float* dst, const float* lds, int count, int lane) {
    int n4 = count >> 2;
    float4* d4 = reinterpret_cast<float4*>(dst);
    const float4* l4 = reinterpret_cast<const float4*>(lds);
#pragma unroll
    for (int j = 0; j < (SH_WAVE_F4 + 63) / 64; j++) {
        int i4 = lane + 64 * j;
        if (i4 < n4) d4[i4] = l4[i4];
    }
    int rem = count & 3;
    if (lane < rem) dst[4 * n4 + lane] = lds[4 * n4 + lane];
}

// ---- training-path backward: ONE camera, SH colours, 16-byte aligned shN / v_shN.
// One thread per Gaussian.  The wave's shN slice comes in through LDS; each coefficient is
// read once and its slot is overwritten in place with its gradient, then the slice leaves as
// v_shN with 16-byte stores: no 45-float register arrays (the generic kernel needs 286 VGPRs,
// i.e. one wave per SIMD; this one fits four).  Also accumulates the densify statistics that
// gsplat's DefaultStrategy._update_state computes in torch.
//
// FUSE = true additionally applies the Adam step (and, when asked, the splatfacto scale
// regulariser) to this Gaussian's 59 parameters right here, where its gradients already sit in
// registers / in the LDS slice: the 944 MB per step that a separate Adam launch spends on
// writing and re-reading the gradients disappear, and the shN group (76 % of all parameters)
// is updated with whole-wave 16-byte accesses.  Parameters are then read and written through
// the same pointers, hence no __restrict__ on them.
// (Round 3, measured and removed: starting the three blocks a CU holds first 6 - 35 us apart, against the idea that the whole
// GPU computes and then streams in lockstep: 541.8 - 555.5 us against 544.4 us without, profiles/r03_bwd_adam_stagger_ab.txt.
// The phases' times add up because each wave's life is a serial chain, not because the waves march in step.)
struct FusedAdam {
    float* m[6];            // means, quats, scales, opacities, sh0, shN
    float* v[6];
    float* sh0;             // parameter itself (the backward does not otherwise read it)
    float step_size[6];     // lr / (1 - beta1^t)
    float b1, b2, eps, inv_bc2_sqrt;
    float sreg_weight, sreg_max_ratio;   // scale regulariser of this step, weight 0 = off
    float mcmc_opacity_reg, mcmc_scale_reg;      // gsplat's MCMC regularisers (every Gaussian, every step), 0 = off
};

// (experiments build only: flags bit 6 / 7 skip the small groups' / the shN Adam -- timing experiments, WRONG results)
#ifdef MI3DGS_EXPERIMENTS
#define MI_BWD_SKIP(flags, bit) ((flags) & (bit))
#else
#define MI_BWD_SKIP(flags, bit) false
#endif
template <bool FUSE>
__global__ __launch_bounds__(256) void project_bwd1_kernel(
    int N, float* means, float* quats, float* scales, float* opacities, float* shN, int sh_degree, FusedAdam A,
    const float* __restrict__ viewmats, const float* __restrict__ Ks, int W, int H, float eps2d, int flags,
    const int32_t* __restrict__ radii, const float* __restrict__ splats, float* v_splats /* read; cleared with MI_FLAG_CLEAR_VSPLATS */,
    float* __restrict__ v_means, float* __restrict__ v_quats, float* __restrict__ v_scales,
    float* __restrict__ v_opacities, float* __restrict__ v_sh0, float* __restrict__ v_shN,
    float* __restrict__ stat_grad2d, float* __restrict__ stat_count, float* __restrict__ stat_radii, int stat_use_abs) {
    __shared__ float4 sSH4[4][SH_WAVE_F4];
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    const int n_raw = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = n_raw < N;
    const int n = live ? n_raw : N - 1;
    const int n0 = n_raw - lane;
    const int count = max(0, min(64, N - n0)) * 45;          // floats in this wave's slice
    float* const slice = reinterpret_cast<float*>(sSH4[wv]);
    float* const my = slice + 45 * lane;

    int2 rad = make_int2(0, 0);
    if (live) rad = *reinterpret_cast<const int2*>(radii + 2 * (long long)n);
    const bool vis = live && rad.x > 0 && rad.y > 0;
    const bool any_vis = wave_ballot(vis) != 0ull;
    if (FUSE && (((flags & MI_FLAG_ONLY_CULLED_WAVES) && any_vis) || ((flags & MI_FLAG_ONLY_VISIBLE_WAVES) && !any_vis))) return;
    const bool need_coef = sh_degree >= 1 && any_vis;
    // FUSE: the lane that stages float4 number i4 of the slice is also the lane that applies Adam to
    // it at the end, so the coefficients stay in 48 registers instead of being read from HBM twice
    // (the kernel sits at 3 waves per SIMD because of its LDS, which leaves 168 VGPRs).
    float4 keep[(SH_WAVE_F4 + 63) / 64];
    if (need_coef) {
        if (FUSE) {
            const float4* s4 = reinterpret_cast<const float4*>(shN + 45 * (long long)n0);
            float4* l4 = reinterpret_cast<float4*>(slice);
            const int n4s = count >> 2;                      // >= 11: a wave holds at least one Gaussian
            // Loads first, unconditionally (index clamped), LDS stores after.  With the bounds test around
            // each load the compiler emitted load -> s_waitcnt vmcnt(0) -> ds_write twelve times in a row:
            // twelve memory latencies in series at the head of every wave.
#pragma unroll
            for (int j = 0; j < (SH_WAVE_F4 + 63) / 64; j++) keep[j] = s4[min(lane + 64 * j, n4s - 1)];
#pragma unroll
            for (int j = 0; j < (SH_WAVE_F4 + 63) / 64; j++) {
                int i4 = lane + 64 * j;
                if (i4 < n4s) l4[i4] = keep[j];
            }
            if (lane < (count & 3)) slice[(count & ~3) + lane] = shN[45 * (long long)n0 + (count & ~3) + lane];
        } else {
            slice_load(slice, shN + 45 * (long long)n0, count, lane);
        }
    }
    // same-wave LDS hand-off: ordering only.  Wavefront scope: a workgroup-scope fence also waits for every global access in
    // flight (s_waitcnt vmcnt(0)), which put a full memory round trip between the slice and the loads behind it.
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();

    GeoGrad G;
#pragma unroll
    for (int i = 0; i < 3; i++) G.vmean[i] = 0.f;
#pragma unroll
    for (int i = 0; i < 9; i++) G.vSw[i] = 0.f;
    G.vopa = 0.f;
    float vc0[3] = {0.f, 0.f, 0.f};
    float g2d = 0.f, cnt = 0.f, rmax = 0.f;
    float q[4] = {1.f, 0.f, 0.f, 0.f}, s[3] = {1.f, 1.f, 1.f}, Rq[9], inv_norm = 1.f, opa_act = 1.f;

    if (vis) {
        float mean[3] = {means[3 * n], means[3 * n + 1], means[3 * n + 2]};
        float4 q4 = *reinterpret_cast<const float4*>(quats + 4 * (long long)n);
        q[0] = q4.x; q[1] = q4.y; q[2] = q4.z; q[3] = q4.w;
        s[0] = scales[3 * n]; s[1] = scales[3 * n + 1]; s[2] = scales[3 * n + 2];
        if (flags & MI_FLAG_LOG_SCALES) { s[0] = __expf(s[0]); s[1] = __expf(s[1]); s[2] = __expf(s[2]); }
        float opa_raw = opacities ? opacities[n] : 1.f;
        opa_act = (flags & MI_FLAG_LOGIT_OPAC) ? sigmoidf(opa_raw) : opa_raw;
        float Sw[9];
        quat_to_rotmat(q, Rq, inv_norm);
        covar_world(Rq, s, Sw);
        float4* vr = reinterpret_cast<float4*>(v_splats + (long long)n * GRAD_STRIDE);
        float4 g0 = vr[0], g1 = vr[1], g2 = vr[2];
        if (flags & MI_FLAG_CLEAR_VSPLATS) {        // rasterize_bwd accumulates into floats 0..10 of visible rows only
            const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
            vr[0] = z4; vr[1] = z4; vr[2] = z4;
        }
        float v_rgb[3] = {g1.z, g1.w, g2.x};
        {
            float sx = stat_use_abs ? g2.y : g0.x, sy = stat_use_abs ? g2.z : g0.y;
            sx *= 0.5f * W; sy *= 0.5f * H;
            g2d = sqrtf(sx * sx + sy * sy);
            cnt = 1.f;
            rmax = fmaxf((float)rad.x, (float)rad.y) / (float)max(W, H);
        }
        Cam cam = load_cam(viewmats, Ks, 0);
        const float4* sr = reinterpret_cast<const float4*>(splats + (long long)n * SPLAT_STRIDE);
        const float4 s0 = sr[0], s1 = sr[1], s2 = sr[2];
        geo_bwd_one_camera(cam, mean, Sw, W, H, eps2d, flags, opa_act, g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g2.w, s0.z, s0.w, s1.x, G);
        // ---- colour path
        if (s1.z <= 0.f) v_rgb[0] = 0.f;        // clamp(+0.5, min 0) mask
        if (s1.w <= 0.f) v_rgb[1] = 0.f;
        if (s2.x <= 0.f) v_rgb[2] = 0.f;
        float cp[3];
#pragma unroll
        for (int i = 0; i < 3; i++) cp[i] = -(cam.R[i] * cam.t[0] + cam.R[3 + i] * cam.t[1] + cam.R[6 + i] * cam.t[2]);
        float dx = mean[0] - cp[0], dy = mean[1] - cp[1], dz = mean[2] - cp[2];
        float inv = rsqrtf(fmaxf(dx * dx + dy * dy + dz * dz, 1e-24f));
        float ux = dx * inv, uy = dy * inv, uz = dz * inv;
        float b[16], gk[16];
        sh_basis(sh_degree, ux, uy, uz, b);
        vc0[0] = b[0] * v_rgb[0]; vc0[1] = b[0] * v_rgb[1]; vc0[2] = b[0] * v_rgb[2];
        gk[0] = 0.f;
        const int nb = (sh_degree + 1) * (sh_degree + 1);
#pragma unroll
        for (int k = 1; k < 16; k++) {
            if (k < nb) {
                float c0 = my[3 * (k - 1)], c1 = my[3 * (k - 1) + 1], c2 = my[3 * (k - 1) + 2];
                gk[k] = v_rgb[0] * c0 + v_rgb[1] * c1 + v_rgb[2] * c2;
                my[3 * (k - 1)] = b[k] * v_rgb[0]; my[3 * (k - 1) + 1] = b[k] * v_rgb[1]; my[3 * (k - 1) + 2] = b[k] * v_rgb[2];
            } else {
                gk[k] = 0.f;
                my[3 * (k - 1)] = 0.f; my[3 * (k - 1) + 1] = 0.f; my[3 * (k - 1) + 2] = 0.f;
            }
        }
        float vu[3];
        sh_basis_vjp(sh_degree, ux, uy, uz, gk, vu);
        float dot = vu[0] * ux + vu[1] * uy + vu[2] * uz;       // normalisation vjp
        G.vmean[0] += (vu[0] - dot * ux) * inv;
        G.vmean[1] += (vu[1] - dot * uy) * inv;
        G.vmean[2] += (vu[2] - dot * uz) * inv;
    } else if (live) {
#pragma unroll
        for (int k = 0; k < 45; k++) my[k] = 0.f;
    }
    if (live) {
        float vq[4] = {0.f, 0.f, 0.f, 0.f}, vs[3] = {0.f, 0.f, 0.f};
        if (vis) {
            covar_bwd(Rq, s, q, inv_norm, G.vSw, vq, vs);
            if (flags & MI_FLAG_LOG_SCALES) { vs[0] *= s[0]; vs[1] *= s[1]; vs[2] *= s[2]; }
            if (flags & MI_FLAG_LOGIT_OPAC) G.vopa *= opa_act * (1.f - opa_act);
            drop_nonfinite_geo(G.vmean, vq, vs);
        }
        if (stat_grad2d && vis) {
            stat_grad2d[n] += g2d;
            stat_count[n] += cnt;
            if (stat_radii) stat_radii[n] = fmaxf(stat_radii[n], rmax);
        }
        if (!FUSE) {
            *reinterpret_cast<float4*>(v_quats + 4 * (long long)n) = make_float4(vq[0], vq[1], vq[2], vq[3]);
            v_scales[3 * n] = vs[0]; v_scales[3 * n + 1] = vs[1]; v_scales[3 * n + 2] = vs[2];
            v_means[3 * n] = G.vmean[0]; v_means[3 * n + 1] = G.vmean[1]; v_means[3 * n + 2] = G.vmean[2];
            if (v_opacities) v_opacities[n] = G.vopa;
            v_sh0[3 * n] = vc0[0]; v_sh0[3 * n + 1] = vc0[1]; v_sh0[3 * n + 2] = vc0[2];
        } else if (!MI_BWD_SKIP(flags, 64)) {
            // ---- Adam on the five small groups, one thread per Gaussian
            float sl[3] = {scales[3 * n], scales[3 * n + 1], scales[3 * n + 2]};
            if (A.sreg_weight > 0.f) {
                // splatfacto use_scale_regularization: weight * mean(max(ratio, max_ratio) - max_ratio)
                float mx = fmaxf(sl[0], fmaxf(sl[1], sl[2])), mn = fminf(sl[0], fminf(sl[1], sl[2]));
                float ratio = __expf(mx - mn);
                if (ratio > A.sreg_max_ratio) {
                    float gg = A.sreg_weight * ratio / (float)N;
                    int nmx = (sl[0] == mx) + (sl[1] == mx) + (sl[2] == mx);
                    int nmn = (sl[0] == mn) + (sl[1] == mn) + (sl[2] == mn);
#pragma unroll
                    for (int i = 0; i < 3; i++)
                        vs[i] += (sl[i] == mx ? gg / (float)nmx : 0.f) - (sl[i] == mn ? gg / (float)nmn : 0.f);
                }
            }
// All 42 loads (parameter, exp_avg, exp_avg_sq of the 14 small-group values) first, then the
            // arithmetic, then the stores.  Row by row, the possible aliasing between the parameter and
            // moment pointers kept every row's loads behind the previous row's stores: fourteen more
            // memory latencies in series.
            // (a missing opacity array is read through `means` so that the load section has no branch, and skipped when storing)
            const bool has_opa = opacities != nullptr;
            float* const P_[5] = {means, quats, scales, has_opa ? opacities : means, A.sh0};
            float* const M_[5] = {A.m[0], A.m[1], A.m[2], has_opa ? A.m[3] : means, A.m[4]};
            float* const V_[5] = {A.v[0], A.v[1], A.v[2], has_opa ? A.v[3] : means, A.v[4]};
            constexpr int W_[5] = {3, 4, 3, 1, 3};
            float pp[14], mm[14], vv[14];
            {
                int e = 0;
#pragma unroll
                for (int gi = 0; gi < 5; gi++) {
#pragma unroll
                    for (int i = 0; i < W_[gi]; i++, e++) {
                        long long o = (long long)W_[gi] * n + i;
                        pp[e] = P_[gi][o];
                        mm[e] = M_[gi][o];
                        vv[e] = V_[gi][o];
                    }
                }
            }
            float gsm[14] = {G.vmean[0], G.vmean[1], G.vmean[2], vq[0], vq[1], vq[2], vq[3], vs[0], vs[1], vs[2],
                             G.vopa, vc0[0], vc0[1], vc0[2]};
            if (A.mcmc_opacity_reg != 0.f || A.mcmc_scale_reg != 0.f) {
                // gsplat MCMC (simple_trainer mcmc / splatfacto-mcmc): loss += opacity_reg mean(sigmoid(o)) + scale_reg mean(exp(s)),
                // for EVERY Gaussian -- what mi3dgs_mcmc_regularise adds to the stored gradients of the unfused path
                // (pp[7..9] = log-scales, pp[10] = opacity logit, read above)
                const float o = sigmoidf(pp[10]);
                if (has_opa) gsm[10] += A.mcmc_opacity_reg * o * (1.f - o) / (float)N;
#pragma unroll
                for (int i = 0; i < 3; i++) gsm[7 + i] += A.mcmc_scale_reg * __expf(pp[7 + i]) / (3.f * (float)N);
            }
            {
                int e = 0;
#pragma unroll
                for (int gi = 0; gi < 5; gi++) {
#pragma unroll
                    for (int i = 0; i < W_[gi]; i++, e++)
                        mi_adam1(pp[e], gsm[e], mm[e], vv[e], A.step_size[gi], A.b1, A.b2, A.inv_bc2_sqrt, A.eps);
                }
            }
            {
                int e = 0;
#pragma unroll
                for (int gi = 0; gi < 5; gi++) {
#pragma unroll
                    for (int i = 0; i < W_[gi]; i++, e++) {
                        long long o = (long long)W_[gi] * n + i;
                        if (gi != 3 || has_opa) { P_[gi][o] = pp[e]; M_[gi][o] = mm[e]; V_[gi][o] = vv[e]; }
                    }
                }
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");          // (not workgroup: that waits for the 42 small-group stores)
    __builtin_amdgcn_wave_barrier();
    if (!FUSE) {
        slice_store(v_shN + 45 * (long long)n0, slice, count, lane);
    } else if (!MI_BWD_SKIP(flags, 128) && count > 0) {
        // ---- Adam on the wave's shN slice: gradients from LDS, p / m / v streamed with 16-byte accesses
        // (count > 0: a wave wholly behind the last Gaussian -- the tail of the last block -- has nothing here, and its clamped
        //  load index below would be -1 at an offset past the end of the arrays: reads of up to 45 * 4 * 192 bytes beyond them,
        //  harmless inside a caching allocator's segment and a device fault at a segment's end.  Found late in round 3 by a test
        //  order that put a 333-Gaussian bank at the end of one.)
        const long long off = 45 * (long long)n0;
        const int n4 = count >> 2;
        const float4* g4 = reinterpret_cast<const float4*>(slice);
        float4* p4 = reinterpret_cast<float4*>(shN + off);
        float4* m4 = reinterpret_cast<float4*>(A.m[5] + off);
        float4* v4 = reinterpret_cast<float4*>(A.v[5] + off);
        // moments of four float4s per lane in flight at a time (loads, then arithmetic, then stores; see above)
        constexpr int NJ = (SH_WAVE_F4 + 63) / 64, GJ = 4;
#pragma unroll
        for (int j0 = 0; j0 < NJ; j0 += GJ) {
            float4 mq[GJ], vq4[GJ], pq[GJ];
#pragma unroll
            for (int jj = 0; jj < GJ; jj++) {
                int i4 = min(lane + 64 * (j0 + jj), n4 - 1);
                mq[jj] = m4[i4];
                vq4[jj] = v4[i4];
                pq[jj] = need_coef ? keep[(j0 + jj) < NJ ? (j0 + jj) : 0] : p4[i4];
            }
#pragma unroll
            for (int jj = 0; jj < GJ; jj++) {
                int i4 = lane + 64 * (j0 + jj);
                if (j0 + jj < NJ && i4 < n4) {
                    float4 g = g4[i4], pp = pq[jj], mm = mq[jj], vv = vq4[jj];
                    mi_adam1(pp.x, g.x, mm.x, vv.x, A.step_size[5], A.b1, A.b2, A.inv_bc2_sqrt, A.eps);
                    mi_adam1(pp.y, g.y, mm.y, vv.y, A.step_size[5], A.b1, A.b2, A.inv_bc2_sqrt, A.eps);
                    mi_adam1(pp.z, g.z, mm.z, vv.z, A.step_size[5], A.b1, A.b2, A.inv_bc2_sqrt, A.eps);
                    mi_adam1(pp.w, g.w, mm.w, vv.w, A.step_size[5], A.b1, A.b2, A.inv_bc2_sqrt, A.eps);
                    p4[i4] = pp; m4[i4] = mm; v4[i4] = vv;
                }
            }
        }
        int rem = count & 3;
        if (lane < rem) {
            long long o = off + 4 * n4 + lane;
            float pp = shN[o], mm = A.m[5][o], vv = A.v[5][o];
            mi_adam1(pp, slice[4 * n4 + lane], mm, vv, A.step_size[5], A.b1, A.b2, A.inv_bc2_sqrt, A.eps);
            shN[o] = pp; A.m[5][o] = mm; A.v[5][o] = vv;
        }
    }
}

// ---- generic backward: any number of cameras, any colour mode, any alignment.  One thread per
// Gaussian, loops over cameras and sums in registers; every output is written exactly once.
__global__ __launch_bounds__(256) void project_bwd_kernel(
    int C, int N, const float* __restrict__ means, const float* __restrict__ quats,
    const float* __restrict__ scales, const float* __restrict__ opacities,
    const float* __restrict__ sh0, const float* __restrict__ shN, int color_mode, int sh_degree,
    const float* __restrict__ viewmats, const float* __restrict__ Ks, int W, int H, float eps2d, int flags,
    const int32_t* __restrict__ radii, const float* __restrict__ splats, const float* __restrict__ v_splats,
    float* __restrict__ v_means, float* __restrict__ v_quats, float* __restrict__ v_scales,
    float* __restrict__ v_opacities, float* __restrict__ v_sh0, float* __restrict__ v_shN,
    float* __restrict__ v_colors,
    float* __restrict__ stat_grad2d, float* __restrict__ stat_count, float* __restrict__ stat_radii,
    int stat_use_abs) {
    int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float mean[3] = {means[3 * n], means[3 * n + 1], means[3 * n + 2]};
    float q[4] = {quats[4 * n], quats[4 * n + 1], quats[4 * n + 2], quats[4 * n + 3]};
    float s[3] = {scales[3 * n], scales[3 * n + 1], scales[3 * n + 2]};
    if (flags & MI_FLAG_LOG_SCALES) { s[0] = __expf(s[0]); s[1] = __expf(s[1]); s[2] = __expf(s[2]); }
    float opa_raw = opacities ? opacities[n] : 1.f;
    float opa_act = (flags & MI_FLAG_LOGIT_OPAC) ? sigmoidf(opa_raw) : opa_raw;
    float Rq[9], inv_norm, Sw[9];
    quat_to_rotmat(q, Rq, inv_norm);
    covar_world(Rq, s, Sw);

    GeoGrad G;
#pragma unroll
    for (int i = 0; i < 3; i++) G.vmean[i] = 0.f;
#pragma unroll
    for (int i = 0; i < 9; i++) G.vSw[i] = 0.f;
    G.vopa = 0.f;
    float vc0[3] = {0.f, 0.f, 0.f};
    float vcN[45];
    int nb = (sh_degree + 1) * (sh_degree + 1);
    if (color_mode == 0) {
#pragma unroll
        for (int k = 0; k < 45; k++) vcN[k] = 0.f;
    }
    float g2d = 0.f, cnt = 0.f, rmax = 0.f;

    for (int c = 0; c < C; c++) {
        long long idx = (long long)c * N + n;
        int2 rad = *reinterpret_cast<const int2*>(radii + idx * 2);
        if (rad.x <= 0 || rad.y <= 0) {
            if (color_mode == 2 && v_colors) { v_colors[3 * idx] = 0.f; v_colors[3 * idx + 1] = 0.f; v_colors[3 * idx + 2] = 0.f; }
            continue;
        }
        const float4* vr = reinterpret_cast<const float4*>(v_splats + idx * GRAD_STRIDE);
        float4 g0 = vr[0], g1 = vr[1], g2 = vr[2];
        float v_rgb[3] = {g1.z, g1.w, g2.x};
        {
            float sx = stat_use_abs ? g2.y : g0.x, sy = stat_use_abs ? g2.z : g0.y;
            sx *= 0.5f * W * C; sy *= 0.5f * H * C;
            g2d += sqrtf(sx * sx + sy * sy);
            cnt += 1.f;
            rmax = fmaxf(rmax, fmaxf((float)rad.x, (float)rad.y) / (float)max(W, H));
        }
        Cam cam = load_cam(viewmats, Ks, c);
        const float4* sr = reinterpret_cast<const float4*>(splats + idx * SPLAT_STRIDE);
        const float4 s0 = sr[0], s1 = sr[1];
        geo_bwd_one_camera(cam, mean, Sw, W, H, eps2d, flags, opa_act, g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g2.w, s0.z, s0.w, s1.x, G);
        if (color_mode == 0) {
            float4 s2 = sr[2];
            if (s1.z <= 0.f) v_rgb[0] = 0.f;        // clamp(+0.5, min 0) mask
            if (s1.w <= 0.f) v_rgb[1] = 0.f;
            if (s2.x <= 0.f) v_rgb[2] = 0.f;
            float cp[3];
#pragma unroll
            for (int i = 0; i < 3; i++) cp[i] = -(cam.R[i] * cam.t[0] + cam.R[3 + i] * cam.t[1] + cam.R[6 + i] * cam.t[2]);
            float dx = mean[0] - cp[0], dy = mean[1] - cp[1], dz = mean[2] - cp[2];
            float inv = rsqrtf(fmaxf(dx * dx + dy * dy + dz * dz, 1e-24f));
            float ux = dx * inv, uy = dy * inv, uz = dz * inv;
            float b[16], gk[16];
            sh_basis(sh_degree, ux, uy, uz, b);
            vc0[0] += b[0] * v_rgb[0]; vc0[1] += b[0] * v_rgb[1]; vc0[2] += b[0] * v_rgb[2];
            gk[0] = 0.f;
            const float* cN = shN + 45 * (long long)n;
#pragma unroll
            for (int k = 1; k < 16; k++) {
                if (k < nb) {
                    vcN[3 * (k - 1)] += b[k] * v_rgb[0];
                    vcN[3 * (k - 1) + 1] += b[k] * v_rgb[1];
                    vcN[3 * (k - 1) + 2] += b[k] * v_rgb[2];
                    gk[k] = v_rgb[0] * cN[3 * (k - 1)] + v_rgb[1] * cN[3 * (k - 1) + 1] + v_rgb[2] * cN[3 * (k - 1) + 2];
                } else gk[k] = 0.f;
            }
            float vu[3];
            sh_basis_vjp(sh_degree, ux, uy, uz, gk, vu);
            float dot = vu[0] * ux + vu[1] * uy + vu[2] * uz;   // normalisation vjp
            G.vmean[0] += (vu[0] - dot * ux) * inv;
            G.vmean[1] += (vu[1] - dot * uy) * inv;
            G.vmean[2] += (vu[2] - dot * uz) * inv;
        } else if (v_colors) {
            if (color_mode == 2) { v_colors[3 * idx] = v_rgb[0]; v_colors[3 * idx + 1] = v_rgb[1]; v_colors[3 * idx + 2] = v_rgb[2]; }
            else { vc0[0] += v_rgb[0]; vc0[1] += v_rgb[1]; vc0[2] += v_rgb[2]; }
        }
    }
    float vq[4], vs[3];
    covar_bwd(Rq, s, q, inv_norm, G.vSw, vq, vs);
    if (flags & MI_FLAG_LOG_SCALES) { vs[0] *= s[0]; vs[1] *= s[1]; vs[2] *= s[2]; }
    drop_nonfinite_geo(G.vmean, vq, vs);
#pragma unroll
    for (int i = 0; i < 4; i++) v_quats[4 * n + i] = vq[i];
    v_scales[3 * n] = vs[0]; v_scales[3 * n + 1] = vs[1]; v_scales[3 * n + 2] = vs[2];
    v_means[3 * n] = G.vmean[0]; v_means[3 * n + 1] = G.vmean[1]; v_means[3 * n + 2] = G.vmean[2];
    if (v_opacities) {
        if (flags & MI_FLAG_LOGIT_OPAC) G.vopa *= opa_act * (1.f - opa_act);
        v_opacities[n] = G.vopa;
    }
    if (color_mode == 0) {
        v_sh0[3 * n] = vc0[0]; v_sh0[3 * n + 1] = vc0[1]; v_sh0[3 * n + 2] = vc0[2];
        float* o = v_shN + 45 * (long long)n;
#pragma unroll
        for (int k = 0; k < 45; k++) o[k] = vcN[k];
    } else if (color_mode == 1 && v_colors) {
        v_colors[3 * n] = vc0[0]; v_colors[3 * n + 1] = vc0[1]; v_colors[3 * n + 2] = vc0[2];
    }
    if (stat_grad2d) {
        stat_grad2d[n] += g2d;
        stat_count[n] += cnt;
        if (stat_radii) stat_radii[n] = fmaxf(stat_radii[n], rmax);
    }
}

}  // namespace

extern "C" int mi3dgs_project_fwd(int C, int N, const float* means, const float* quats, const float* scales,
                                  const float* opacities, const float* sh0, const float* shN,
                                  const float* colors, int color_mode, int sh_degree, const float* viewmats,
                                  const float* Ks, int width, int height, float eps2d, float near_plane,
                                  float far_plane, float radius_clip, int flags, int32_t* radii, float* splats,
                                  uint32_t* depth_keys_opt, void* stream) {
    MI_REQUIRE(C > 0 && N >= 0 && width > 0 && height > 0, "project_fwd: bad sizes");
    MI_REQUIRE(color_mode >= 0 && color_mode <= 2, "project_fwd: color_mode must be 0, 1 or 2");
    MI_REQUIRE(color_mode != 0 || (sh_degree >= 0 && sh_degree <= 3 && sh0 && (sh_degree == 0 || shN)),
               "project_fwd: SH mode needs sh0/shN and 0 <= sh_degree <= 3");
    MI_REQUIRE(color_mode == 0 || colors, "project_fwd: colour mode needs colors");
    if (N == 0) return 0;
    long long total = (long long)C * N;
    MI_LAUNCH("project_fwd", project_fwd_kernel, dim3(mi_div_up(total, 256)), dim3(256), 0, (hipStream_t)stream, C, N, means,
                       quats, scales, opacities, sh0, shN, colors, color_mode, sh_degree, viewmats, Ks, width, height,
                       eps2d, near_plane, far_plane, radius_clip, flags, radii, splats, depth_keys_opt);
    MI_LAUNCH_CHECK();
    return 0;
}

extern "C" int mi3dgs_project_bwd(int C, int N, const float* means, const float* quats, const float* scales,
                                  const float* opacities, const float* sh0, const float* shN, int color_mode,
                                  int sh_degree, const float* viewmats, const float* Ks, int width, int height,
                                  float eps2d, int flags, const int32_t* radii, const float* splats,
                                  const float* v_splats, float* v_means, float* v_quats, float* v_scales,
                                  float* v_opacities, float* v_sh0, float* v_shN, float* v_colors,
                                  float* stat_grad2d, float* stat_count, float* stat_radii, int stat_use_abs,
                                  void* stream) {
    MI_REQUIRE(C > 0 && N >= 0, "project_bwd: bad sizes");
    MI_REQUIRE(v_means && v_quats && v_scales, "project_bwd: v_means/v_quats/v_scales required");
    MI_REQUIRE(color_mode != 0 || (v_sh0 && v_shN), "project_bwd: SH mode needs v_sh0/v_shN");
    if (N == 0) return 0;
    const bool fast = C == 1 && color_mode == 0 && ((((uintptr_t)shN) | ((uintptr_t)v_shN) | ((uintptr_t)quats) |
                                                     ((uintptr_t)v_quats)) & 15) == 0;
    MI_REQUIRE(fast || !(flags & MI_FLAG_CLEAR_VSPLATS), "project_bwd: the CLEAR_VSPLATS flag needs the one-camera SH path");
    if (fast) {
        FusedAdam none = {};
        MI_LAUNCH("project_bwd", (project_bwd1_kernel<false>), dim3(mi_div_up(N, 256)), dim3(256), 0, (hipStream_t)stream, N,
                  const_cast<float*>(means), const_cast<float*>(quats), const_cast<float*>(scales),
                  const_cast<float*>(opacities), const_cast<float*>(shN), sh_degree, none, viewmats, Ks, width, height,
                  eps2d, flags, radii, splats, const_cast<float*>(v_splats), v_means, v_quats, v_scales, v_opacities, v_sh0, v_shN,
                  stat_grad2d, stat_count, stat_radii, stat_use_abs);
    }
    else
        MI_LAUNCH("project_bwd", project_bwd_kernel, dim3(mi_div_up(N, 256)), dim3(256), 0, (hipStream_t)stream, C, N,
                  means, quats, scales, opacities, sh0, shN, color_mode, sh_degree, viewmats, Ks, width, height, eps2d,
                  flags, radii, splats, v_splats, v_means, v_quats, v_scales, v_opacities, v_sh0, v_shN, v_colors,
                  stat_grad2d, stat_count, stat_radii, stat_use_abs);
    MI_LAUNCH_CHECK();
    return 0;
}

// Single-camera SH backward with the Adam step fused in (see project_bwd1_kernel).  The parameter
// arrays are updated IN PLACE; no gradient is written.  exp_avg / exp_avg_sq / lrs follow the
// group order means[3], quats[4], scales[3], opacities[1], sh0[3], shN[45] (HOST arrays of 6).
// scale_reg_weight > 0 adds the splatfacto scale regulariser's gradient for this step.
static int project_bwd_adam_impl(int N, float* means, float* quats, float* scales, float* opacities, float* sh0,
                                 float* shN, int sh_degree, const float* viewmats, const float* Ks, int width,
                                 int height, float eps2d, int flags, const int32_t* radii, const float* splats,
                                 const float* v_splats, float* const* exp_avg, float* const* exp_avg_sq,
                                 const float* lrs, int step, float beta1, float beta2, float eps,
                                 float scale_reg_weight, float scale_reg_max_ratio, float mcmc_opacity_reg, float mcmc_scale_reg,
                                 float* stat_grad2d, float* stat_count, float* stat_radii, int stat_use_abs, void* stream) {
    MI_REQUIRE(N >= 0 && step >= 1, "project_bwd_adam: bad N / step (step is 1-based)");
    MI_REQUIRE(means && quats && scales && sh0 && shN && exp_avg && exp_avg_sq && lrs, "project_bwd_adam: null argument");
    if (N == 0) return 0;
    FusedAdam A;
    uintptr_t align = ((uintptr_t)shN) | ((uintptr_t)quats);
    for (int g = 0; g < 6; g++) {
        A.m[g] = exp_avg[g]; A.v[g] = exp_avg_sq[g];
        MI_REQUIRE(A.m[g] && A.v[g], "project_bwd_adam: null Adam moment buffer");
    }
    align |= ((uintptr_t)A.m[5]) | ((uintptr_t)A.v[5]);
    MI_REQUIRE((align & 15) == 0, "project_bwd_adam: quats, shN and the shN moments must be 16-byte aligned");
    double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    for (int g = 0; g < 6; g++) A.step_size[g] = (float)(lrs[g] / bc1);
    A.sh0 = sh0;
    A.b1 = beta1; A.b2 = beta2; A.eps = eps; A.inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
    A.sreg_weight = scale_reg_weight; A.sreg_max_ratio = scale_reg_max_ratio;
    A.mcmc_opacity_reg = mcmc_opacity_reg; A.mcmc_scale_reg = mcmc_scale_reg;
    // (profiler tag of its own for the every-tenth step that carries splatfacto's scale regulariser)
    MI_LAUNCH(scale_reg_weight > 0.f ? "project_bwd_adam/scale_reg" : "project_bwd_adam", (project_bwd1_kernel<true>), dim3(mi_div_up(N, 256)), dim3(256), 0,
              (hipStream_t)stream, N, means, quats, scales, opacities, shN, sh_degree, A, viewmats, Ks, width, height,
                  eps2d, flags, radii, splats, const_cast<float*>(v_splats), nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, stat_grad2d,
                  stat_count, stat_radii, stat_use_abs);
    MI_LAUNCH_CHECK();
    return 0;
}

extern "C" int mi3dgs_project_bwd_adam(int N, float* means, float* quats, float* scales, float* opacities, float* sh0,
                                       float* shN, int sh_degree, const float* viewmats, const float* Ks, int width,
                                       int height, float eps2d, int flags, const int32_t* radii, const float* splats,
                                       const float* v_splats, float* const* exp_avg, float* const* exp_avg_sq,
                                       const float* lrs, int step, float beta1, float beta2, float eps,
                                       float scale_reg_weight, float scale_reg_max_ratio, float* stat_grad2d,
                                       float* stat_count, float* stat_radii, int stat_use_abs, void* stream) {
    return project_bwd_adam_impl(N, means, quats, scales, opacities, sh0, shN, sh_degree, viewmats, Ks, width, height, eps2d, flags, radii,
                                 splats, v_splats, exp_avg, exp_avg_sq, lrs, step, beta1, beta2, eps, scale_reg_weight,
                                 scale_reg_max_ratio, 0.f, 0.f, stat_grad2d, stat_count, stat_radii, stat_use_abs, stream);
}

// The same with gsplat's MCMC regularisers folded in (opacity_reg * mean(sigmoid(o)) + scale_reg * mean(exp(s)) over every Gaussian):
// the MCMC strategy's training step through the fused kernel instead of backward + regulariser + Adam launches.
extern "C" int mi3dgs_project_bwd_adam_mcmc(int N, float* means, float* quats, float* scales, float* opacities, float* sh0,
                                            float* shN, int sh_degree, const float* viewmats, const float* Ks, int width,
                                            int height, float eps2d, int flags, const int32_t* radii, const float* splats,
                                            const float* v_splats, float* const* exp_avg, float* const* exp_avg_sq,
                                            const float* lrs, int step, float beta1, float beta2, float eps,
                                            float mcmc_opacity_reg, float mcmc_scale_reg, void* stream) {
    MI_REQUIRE(!(flags & (MI_FLAG_ONLY_CULLED_WAVES | MI_FLAG_ONLY_VISIBLE_WAVES)), "project_bwd_adam_mcmc: every Gaussian has a gradient");
    return project_bwd_adam_impl(N, means, quats, scales, opacities, sh0, shN, sh_degree, viewmats, Ks, width, height, eps2d, flags, radii,
                                 splats, v_splats, exp_avg, exp_avg_sq, lrs, step, beta1, beta2, eps, 0.f, 1.f, mcmc_opacity_reg,
                                 mcmc_scale_reg, nullptr, nullptr, nullptr, 0, stream);
}
