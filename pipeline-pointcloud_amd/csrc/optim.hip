// Fused Adam over all Gaussian parameter groups in ONE launch, plus the splatfacto scale
// regulariser.  gfx950 only.
//
// Replaces torch.optim.Adam (one foreach launch set per group in splatfacto / gsplat's
// simple_trainer) reached by the reference through main.py:1312 / main.py:1343, and the
// `use_scale_regularization=True` term the reference passes at main.py:1288
// (SURVEY.md 8a rows a11-a12).
//
// Bound: pure HBM stream, 28 B per parameter float (read p,g,m,v; write p,m,v)
// = 1652 B per Gaussian at 59 floats.
#include "common.h"
#include <math.h>

#define MI_ADAM_MAX_SEGS 8

struct AdamSeg {
    float* p;
    const float* g;
    float* m;
    float* v;
    long long n;       // floats in this segment
    float lr;
    int first_block;   // filled by the host wrapper
};

struct AdamArgs {
    AdamSeg seg[MI_ADAM_MAX_SEGS];
    int nseg;
};

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int AD_THREADS = 256;
constexpr int AD_PER_BLOCK = AD_THREADS * 4 * 4;   // 4 float4 per thread

#define adam1 mi_adam1

// gscale: every gradient is multiplied by it on the way in (the data-parallel trainer's 1 / world: its reduce-scatter
// delivers the SUM over the ranks, and a separate mul_ pass over the slice was a read and a write of it per step)
__global__ __launch_bounds__(AD_THREADS) void adam_kernel(AdamArgs a, float b1, float b2, float eps, float inv_bc1,
                                                          float inv_bc2_sqrt, float gscale) {
    int blk = blockIdx.x;
    int s = 0;
#pragma unroll
    for (int i = 1; i < MI_ADAM_MAX_SEGS; i++)
        if (i < a.nseg && blk >= a.seg[i].first_block) s = i;
    const AdamSeg sg = a.seg[s];
    long long base = (long long)(blk - sg.first_block) * AD_PER_BLOCK;
    float step_size = sg.lr * inv_bc1;
    bool aligned = ((((uintptr_t)sg.p) | ((uintptr_t)sg.g) | ((uintptr_t)sg.m) | ((uintptr_t)sg.v)) & 15) == 0;
    // Fast path (whole block inside an aligned segment): the 16 loads of a thread go out together, then
    // the arithmetic, then the stores.  Iteration by iteration the stores of one round kept the loads
    // of the next behind them (the four pointers may alias as far as the compiler knows).
    if (aligned && base + AD_PER_BLOCK <= sg.n) {
        f32x4 p[4], g[4], m[4], v[4];
#pragma unroll
        for (int it = 0; it < 4; it++) {
            long long i = base + ((long long)it * AD_THREADS + threadIdx.x) * 4;
            // streamed once per step, never re-read before the next step: non-temporal
            p[it] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(sg.p + i));
            g[it] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(sg.g + i));
            m[it] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(sg.m + i));
            v[it] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(sg.v + i));
        }
#pragma unroll
        for (int it = 0; it < 4; it++) {
            long long i = base + ((long long)it * AD_THREADS + threadIdx.x) * 4;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                float pe = p[it][e], me = m[it][e], ve = v[it][e];
                adam1(pe, g[it][e] * gscale, me, ve, step_size, b1, b2, inv_bc2_sqrt, eps);
                p[it][e] = pe; m[it][e] = me; v[it][e] = ve;
            }
            *reinterpret_cast<f32x4*>(sg.p + i) = p[it];     // parameters are re-read by the next forward
            __builtin_nontemporal_store(m[it], reinterpret_cast<f32x4*>(sg.m + i));
            __builtin_nontemporal_store(v[it], reinterpret_cast<f32x4*>(sg.v + i));
        }
        return;
    }
#pragma unroll
    for (int it = 0; it < 4; it++) {
        long long i = base + ((long long)it * AD_THREADS + threadIdx.x) * 4;
        if (i >= sg.n) break;
        for (int k = 0; k < 4 && i + k < sg.n; k++) {
            float p = sg.p[i + k], m = sg.m[i + k], v = sg.v[i + k];
            adam1(p, sg.g[i + k] * gscale, m, v, step_size, b1, b2, inv_bc2_sqrt, eps);
            sg.p[i + k] = p; sg.m[i + k] = m; sg.v[i + k] = v;
        }
    }
}

// loss += weight * mean(max(smax/smin, max_ratio) - max_ratio); v_scales (log-space) += d/ds
__global__ __launch_bounds__(256) void scale_reg_kernel(int N, const float* __restrict__ scales_log, float weight,
                                                        float max_ratio, float* __restrict__ v_scales,
                                                        float* __restrict__ loss_sum) {
    int n = blockIdx.x * blockDim.x + threadIdx.x;
    float contrib = 0.f;
    if (n < N) {
        float s0 = scales_log[3 * n], s1 = scales_log[3 * n + 1], s2 = scales_log[3 * n + 2];
        float mx = fmaxf(s0, fmaxf(s1, s2)), mn = fminf(s0, fminf(s1, s2));
        float ratio = __expf(mx - mn);
        if (ratio > max_ratio) {
            contrib = ratio - max_ratio;
            if (v_scales) {
                float g = weight * ratio / (float)N;
                // torch amax/amin backward: gradient split evenly between ties
                int nmx = (s0 == mx) + (s1 == mx) + (s2 == mx);
                int nmn = (s0 == mn) + (s1 == mn) + (s2 == mn);
                float gm = g / (float)nmx, gn = g / (float)nmn;
                v_scales[3 * n] += (s0 == mx ? gm : 0.f) - (s0 == mn ? gn : 0.f);
                v_scales[3 * n + 1] += (s1 == mx ? gm : 0.f) - (s1 == mn ? gn : 0.f);
                v_scales[3 * n + 2] += (s2 == mx ? gm : 0.f) - (s2 == mn ? gn : 0.f);
            }
        }
    }
    if (loss_sum) {
        contrib = wave_sum_all(contrib);
        if (lane_id() == 0 && contrib != 0.f) atomicAdd(loss_sum, contrib * weight / (float)N);
    }
}

}  // namespace

// One Adam step (torch.optim.Adam semantics, no weight decay / amsgrad) over up to 8 flat
// float segments.  `step` is the 1-based step count after increment.
extern "C" int mi3dgs_adam_step(int nseg, float* const* params, const float* const* grads, float* const* exp_avg,
                                float* const* exp_avg_sq, const long long* numel, const float* lrs, int step,
                                float beta1, float beta2, float eps, float grad_scale, void* stream) {
    MI_REQUIRE(nseg >= 1 && nseg <= MI_ADAM_MAX_SEGS, "adam_step: 1..8 segments");
    MI_REQUIRE(step >= 1, "adam_step: step is 1-based");
    AdamArgs a;
    a.nseg = nseg;
    int blocks = 0;
    for (int i = 0; i < MI_ADAM_MAX_SEGS; i++) {
        if (i < nseg) {
            MI_REQUIRE(numel[i] >= 0, "adam_step: negative numel");
            a.seg[i] = AdamSeg{params[i], grads[i], exp_avg[i], exp_avg_sq[i], numel[i], lrs[i], blocks};
            blocks += mi_div_up(numel[i], AD_PER_BLOCK);
        } else {
            a.seg[i] = AdamSeg{nullptr, nullptr, nullptr, nullptr, 0, 0.f, 0x7fffffff};
        }
    }
    if (blocks == 0) return 0;
    double bc1 = 1.0 - pow((double)beta1, (double)step);
    double bc2 = 1.0 - pow((double)beta2, (double)step);
    MI_LAUNCH("adam", adam_kernel, dim3(blocks), dim3(AD_THREADS), 0, (hipStream_t)stream, a, beta1, beta2, eps,
                       (float)(1.0 / bc1), (float)(1.0 / sqrt(bc2)), grad_scale);
    MI_LAUNCH_CHECK();
    return 0;
}

// ---- Adam of the 64-Gaussian groups none of whose members is visible.  Their gradient is zero, so their update needs nothing
// but mi3dgs_project_fwd's radii: a pure stream over parameters and moments.  As a kernel of its own -- thirty-odd registers,
// no LDS -- it fits on SIMDs beside the forward rasteriser's and the loss kernels' waves (which leave HBM idle), on a second
// stream; mi3dgs_project_bwd_adam with MI3DGS_FLAG_ONLY_VISIBLE_GROUPS then handles the other groups after rasterize_bwd.
// The arithmetic is the fused kernel's for a Gaussian it finds invisible: mi_adam1 with a zero gradient and the same per-step
// scalars (lr / (1 - b1^t) formed in double on the host), so every Gaussian gets the same bits whichever kernel updates it.
namespace {
struct AdamGroupsArgs {
    float* p[6]; float* m[6]; float* v[6];
    float step_size[6];
    float b1, b2, eps, inv_bc2_sqrt, zero;
    float sreg_weight, sreg_max_ratio;      // splatfacto's scale regulariser of this step (weight 0 = off): the one gradient a culled Gaussian has
};

__global__ __launch_bounds__(256) void adam_culled_groups_kernel(int N, const int32_t* __restrict__ radii, AdamGroupsArgs A) {
    const int lane = lane_id();
    const long long n0 = ((long long)blockIdx.x * 256 + threadIdx.x - lane);          // first Gaussian of this wave's group
    if (n0 >= N) return;
    const int cnt = (int)min((long long)64, (long long)N - n0);
    bool vis = false;
    if (lane < cnt) {
        const int2 r = *reinterpret_cast<const int2*>(radii + 2 * (n0 + lane));
        vis = r.x > 0 && r.y > 0;
    }
    if (wave_ballot(vis) != 0ull) return;
    // (the loop over the groups is NOT unrolled: the kernel should stay small enough in registers to sit beside other kernels)
    const int W_[6] = {3, 4, 3, 1, 3, 45};
#pragma unroll 1
    for (int gi = 0; gi < 6; gi++) {
        if (A.p[gi] == nullptr) continue;                     // (no opacity array)
        if (gi == 2 && A.sreg_weight > 0.f) {
            // the scale regulariser's gradient, expression for expression as project_bwd1_kernel<true> forms it (csrc/project.hip),
            // then the same mi_adam1: a lane per Gaussian for this group in such a step (every tenth of `ns-train splatfacto`)
            if (lane < cnt) {
                const long long o = 3 * (n0 + lane);
                float sl[3] = {A.p[2][o], A.p[2][o + 1], A.p[2][o + 2]};
                float mm[3] = {A.m[2][o], A.m[2][o + 1], A.m[2][o + 2]};
                float vv[3] = {A.v[2][o], A.v[2][o + 1], A.v[2][o + 2]};
                float vs[3] = {A.zero, A.zero, A.zero};
                const float mx = fmaxf(sl[0], fmaxf(sl[1], sl[2])), mn = fminf(sl[0], fminf(sl[1], sl[2]));
                const float ratio = __expf(mx - mn);
                if (ratio > A.sreg_max_ratio) {
                    const float gg = A.sreg_weight * ratio / (float)N;
                    const int nmx = (sl[0] == mx) + (sl[1] == mx) + (sl[2] == mx);
                    const int nmn = (sl[0] == mn) + (sl[1] == mn) + (sl[2] == mn);
#pragma unroll
                    for (int i = 0; i < 3; i++)
                        vs[i] += (sl[i] == mx ? gg / (float)nmx : 0.f) - (sl[i] == mn ? gg / (float)nmn : 0.f);
                }
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    mi_adam1(sl[i], vs[i], mm[i], vv[i], A.step_size[2], A.b1, A.b2, A.inv_bc2_sqrt, A.eps);
                    A.p[2][o + i] = sl[i]; A.m[2][o + i] = mm[i]; A.v[2][o + i] = vv[i];
                }
            }
            continue;
        }
        const int total = cnt * W_[gi];
        const long long base = n0 * W_[gi];                  // multiple of 4 floats: 64 Gaussians x any width
        float4* p4 = reinterpret_cast<float4*>(A.p[gi] + base);
        float4* m4 = reinterpret_cast<float4*>(A.m[gi] + base);
        float4* v4 = reinterpret_cast<float4*>(A.v[gi] + base);
        const int n4 = total >> 2;
        const float ss = A.step_size[gi];
        // (one float4 of each array per lane at a time: the kernel's job is to fit on SIMDs whose registers are nearly all taken;
        //  the bandwidth comes from the number of waves that do)
#pragma unroll 1
        for (int i4 = lane; i4 < n4; i4 += 64) {
            float4 pa = p4[i4], ma = m4[i4], va = v4[i4];
            mi_adam1(pa.x, A.zero, ma.x, va.x, ss, A.b1, A.b2, A.inv_bc2_sqrt, A.eps);
            mi_adam1(pa.y, A.zero, ma.y, va.y, ss, A.b1, A.b2, A.inv_bc2_sqrt, A.eps);
            mi_adam1(pa.z, A.zero, ma.z, va.z, ss, A.b1, A.b2, A.inv_bc2_sqrt, A.eps);
            mi_adam1(pa.w, A.zero, ma.w, va.w, ss, A.b1, A.b2, A.inv_bc2_sqrt, A.eps);
            p4[i4] = pa; m4[i4] = ma; v4[i4] = va;
        }
        for (int e = 4 * n4 + lane; e < total; e += 64) {      // (a last group of fewer than 64 Gaussians)
            float pp = A.p[gi][base + e], mm = A.m[gi][base + e], vv = A.v[gi][base + e];
            mi_adam1(pp, A.zero, mm, vv, ss, A.b1, A.b2, A.inv_bc2_sqrt, A.eps);
            A.p[gi][base + e] = pp; A.m[gi][base + e] = mm; A.v[gi][base + e] = vv;
        }
    }
}
}  // namespace

// params / exp_avg / exp_avg_sq: HOST arrays of 6 device pointers in the group order of mi3dgs_project_bwd_adam (opacities may be
// null in all three); radii[N][2] of this step's mi3dgs_project_fwd (one camera).
extern "C" int mi3dgs_adam_culled_groups(int N, float* const* params, float* const* exp_avg, float* const* exp_avg_sq,
                                         const int32_t* radii, const float* lrs, int step, float beta1, float beta2, float eps,
                                         float scale_reg_weight, float scale_reg_max_ratio, void* stream) {
    MI_REQUIRE(N >= 0 && params && exp_avg && exp_avg_sq && radii && lrs, "adam_culled_groups: null argument");
    MI_REQUIRE(step >= 1, "adam_culled_groups: step is 1-based");
    if (N == 0) return 0;
    AdamGroupsArgs A;
    uintptr_t align = 0;
    for (int g = 0; g < 6; g++) {
        A.p[g] = params[g]; A.m[g] = exp_avg[g]; A.v[g] = exp_avg_sq[g];
        MI_REQUIRE((params[g] != nullptr) == (exp_avg[g] != nullptr) && (params[g] != nullptr) == (exp_avg_sq[g] != nullptr),
                   "adam_culled_groups: a group needs all three arrays or none");
        MI_REQUIRE(params[g] != nullptr || g == 3, "adam_culled_groups: only the opacity group may be absent");
        align |= (uintptr_t)params[g] | (uintptr_t)exp_avg[g] | (uintptr_t)exp_avg_sq[g];
    }
    MI_REQUIRE((align & 15) == 0, "adam_culled_groups: arrays must be 16-byte aligned");
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    for (int g = 0; g < 6; g++) A.step_size[g] = (float)(lrs[g] / bc1);
    A.b1 = beta1; A.b2 = beta2; A.eps = eps; A.inv_bc2_sqrt = (float)(1.0 / sqrt(bc2)); A.zero = 0.f;
    A.sreg_weight = scale_reg_weight; A.sreg_max_ratio = scale_reg_max_ratio;
    MI_LAUNCH("adam_culled_groups", adam_culled_groups_kernel, dim3(mi_div_up(N, 256)), dim3(256), 0, (hipStream_t)stream, N, radii, A);
    MI_LAUNCH_CHECK();
    return 0;
}

// ---- HBM yardstick.  What a plain 16-byte-per-lane stream gets from THIS device, measured by the bench itself
// (bench.py: roofline.copy_GBps_this_box) instead of being assumed: NR arrays are read, their sum is written to NW arrays;
// every array is `n4` float4 long and lives in `buf` one after the other (reads first).  (1, 1) is the float4 copy of
// MI355X_MICROARCH.md (6.29 TB/s there); (5, 4) is the read : write mix of project_bwd_adam (1.78 : 1.45 GB).
namespace {
template <int NR, int NW>
__global__ __launch_bounds__(256) void hbm_stream_kernel(float4* __restrict__ buf, long long n4) {
    constexpr int UN = 4;
    const long long base = ((long long)blockIdx.x * UN) * 256 + threadIdx.x;
    float4 acc[UN];
#pragma unroll
    for (int u = 0; u < UN; u++) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int r = 0; r < NR; r++) {
        float4 t[UN];
#pragma unroll
        for (int u = 0; u < UN; u++) {
            const long long i = base + (long long)u * 256;
            t[u] = buf[(long long)r * n4 + (i < n4 ? i : n4 - 1)];
        }
#pragma unroll
        for (int u = 0; u < UN; u++) { acc[u].x += t[u].x; acc[u].y += t[u].y; acc[u].z += t[u].z; acc[u].w += t[u].w; }
    }
    if (NW == 0) {          // read-only: keep the loads alive without storing anything that matters
        float s = 0.f;
#pragma unroll
        for (int u = 0; u < UN; u++) s += acc[u].x + acc[u].y + acc[u].z + acc[u].w;
        if (s == 1.2345e-30f) buf[0].x = s;
    }
#pragma unroll
    for (int w = 0; w < NW; w++)
#pragma unroll
        for (int u = 0; u < UN; u++) {
            const long long i = base + (long long)u * 256;
            if (i < n4) buf[(long long)(NR + w) * n4 + i] = acc[u];
        }
}
}  // namespace

extern "C" int mi3dgs_debug_hbm_stream(float* buf, long long floats_per_array, int n_read, int n_write, void* stream) {
    MI_REQUIRE(buf && floats_per_array > 0 && (floats_per_array & 3) == 0 && (((uintptr_t)buf) & 15) == 0, "hbm_stream: bad buffer");
    const long long n4 = floats_per_array / 4;
    const dim3 grid((unsigned)mi_div_up(n4, 256 * 4)), block(256);
    hipStream_t st = (hipStream_t)stream;
    float4* b4 = reinterpret_cast<float4*>(buf);
#define HS(R, W) else if (n_read == R && n_write == W) MI_LAUNCH("hbm_stream", (hbm_stream_kernel<R, W>), grid, block, 0, st, b4, n4)
    if (false) {}
    HS(1, 1); HS(2, 1); HS(5, 4); HS(1, 0); HS(0, 1); HS(4, 3);
    else MI_REQUIRE(false, "hbm_stream: (n_read, n_write) must be one of (1,1) (2,1) (5,4) (4,3) (1,0) (0,1)");
#undef HS
    MI_LAUNCH_CHECK();
    return 0;
}

extern "C" int mi3dgs_scale_reg(int N, const float* scales_log, float weight, float max_ratio, float* v_scales,
                                float* loss_sum, void* stream) {
    if (N <= 0) return 0;
    MI_LAUNCH("scale_reg", scale_reg_kernel, dim3(mi_div_up(N, 256)), dim3(256), 0, (hipStream_t)stream, N, scales_log,
                       weight, max_ratio, v_scales, loss_sum);
    MI_LAUNCH_CHECK();
    return 0;
}
