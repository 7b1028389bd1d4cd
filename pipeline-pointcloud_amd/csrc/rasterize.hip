// C-ABI entry points of the tile rasteriser (mi3dgs_rasterize_fwd / mi3dgs_rasterize_bwd), and -- in the EXPERIMENTS build
// only -- the round-1 VALU kernels they used to launch.  gfx950 only.
//
// Replaces gsplat rasterize_to_pixels_{fwd,bwd}, reached by the
// reference only through main.py:1312 / main.py:1343.
//
// The kernels are rasterize_mfma.hip (forward) and rasterize_bwd_mm.hip (backward).  (The round-1 VALU kernels that lived here
// until round 3 are gone; their measurements are in docs/FINDINGS_r01_r02.md.)
#include "common.h"

constexpr int TILE = 16;
[[maybe_unused]] constexpr int BLOCK = TILE * TILE;


int mi_rasterize_fwd_mfma(int n_tiles, int width, int height, int tile_width, int tile_height, const float* splats,
                          const int32_t* isect_offsets, const int32_t* flatten_ids, const int32_t* n_isect_dev,
                          const float* backgrounds, float* render, float* alphas, int32_t* last_ids, void* seg_ws, size_t seg_ws_bytes,
                          hipStream_t st);
int mi_rasterize_bwd_mfma(int n_tiles, int width, int height, int tile_width, int tile_height, const float* splats,
                          const int32_t* isect_offsets, const int32_t* flatten_ids, const int32_t* n_isect_dev,
                          const float* backgrounds, const float* alphas, const int32_t* last_ids, const float* v_render,
                          const float* v_alphas, int absgrad, float* v_splats, int variant, long long n_gauss, const float* render,
                          void* seg_ws, size_t seg_ws_bytes, hipStream_t st);
static int g_raster_mode = 1;
// Mode 1 = the product kernels, the only mode the product library has.  The experiments build (libmi3dgs_exp.so) adds:
// 3 = MFMA forward + backward with the all-f32 cross-lane reduce-scatter instead of the bf16 MFMA contraction (correct; the f32
// yardstick of tests/test_gpu_configs.py and of the precision A/B); 4 = backward with THREE-term bf16 pixel sums (24 significant
// bits).  Forward and backward must run in the same mode.
extern "C" int mi3dgs_debug_set_raster_mode(int mode) {
#ifdef MI3DGS_EXPERIMENTS
    MI_REQUIRE(mode == 1 || mode == 3 || mode == 4,
               "set_raster_mode: unknown mode");
    g_raster_mode = mode;
    return 0;
#else
    MI_REQUIRE(mode == 1, "set_raster_mode: this is the product library, it has mode 1 only (experiments: libmi3dgs_exp.so)");
    return 0;
#endif
}

extern "C" int mi3dgs_rasterize_fwd(int C, int width, int height, int tile_size, int tile_width, int tile_height,
                                    const float* splats, const int32_t* isect_offsets, const int32_t* flatten_ids,
                                    const int32_t* n_isect_dev, const float* backgrounds, float* render,
                                    float* alphas, int32_t* last_ids, void* seg_ws, size_t seg_ws_bytes, void* stream) {
    MI_REQUIRE(tile_size == TILE, "rasterize_fwd: tile_size must be 16");
    MI_REQUIRE(C > 0 && width > 0 && height > 0, "rasterize_fwd: bad sizes");
    MI_REQUIRE(tile_width == mi_div_up(width, TILE) && tile_height == mi_div_up(height, TILE),
               "rasterize_fwd: tile grid does not match image size");
    int n_tiles = C * tile_width * tile_height;
    hipStream_t st = (hipStream_t)stream;
    return mi_rasterize_fwd_mfma(n_tiles, width, height, tile_width, tile_height, splats, isect_offsets, flatten_ids,
                                 n_isect_dev, backgrounds, render, alphas, last_ids, seg_ws, seg_ws_bytes, st);
}

extern "C" int mi3dgs_rasterize_bwd(int C, int width, int height, int tile_size, int tile_width, int tile_height,
                                    const float* splats, const int32_t* isect_offsets, const int32_t* flatten_ids,
                                    const int32_t* n_isect_dev, const float* backgrounds, const float* alphas,
                                    const int32_t* last_ids, const float* v_render, const float* v_alphas,
                                    int absgrad, float* v_splats, long long n_gaussians, const float* render, void* seg_ws,
                                    size_t seg_ws_bytes, void* stream) {
    MI_REQUIRE(tile_size == TILE, "rasterize_bwd: tile_size must be 16");
    MI_REQUIRE(tile_width == mi_div_up(width, TILE) && tile_height == mi_div_up(height, TILE),
               "rasterize_bwd: tile grid does not match image size");
    int n_tiles = C * tile_width * tile_height;
    hipStream_t st = (hipStream_t)stream;
    return mi_rasterize_bwd_mfma(n_tiles, width, height, tile_width, tile_height, splats, isect_offsets, flatten_ids,
                                 n_isect_dev, backgrounds, alphas, last_ids, v_render, v_alphas, absgrad, v_splats, g_raster_mode,
                                 n_gaussians, render, seg_ws, seg_ws_bytes, st);
}
