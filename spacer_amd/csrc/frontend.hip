// Video front end on the GPU: bicubic + antialias resize of uint8 frames (the qwen_vl_utils step the reference runs on the
// CPU through torchvision: vision_process.py:310-315, `resize(video, [h, w], BICUBIC, antialias=True)` followed by .float()).
//
// Arithmetic = torch's `_upsample_bicubic2d_aa` (the operator torchvision's tensor resize calls): a separable filter with
// per-output-index windows [xmin, xmin + xsize) and normalised cubic weights (a = -0.5, support 2 * max(scale, 1)), the
// HORIZONTAL pass first, fp32 intermediate, then the vertical pass; torchvision then rounds to the uint8 grid (round half to
// even) and clamps to [0, 255].  The window / weight tables are built by the caller on the host with the same fp32 formulas
// (spacer_amd/qwen_vl_utils/vision_process.py:aa_tables); this file only applies them -- HBM-bound byte work, one thread per
// output element, the weight rows (<= 32 taps) are read through the scalar/L1 caches.
#include "common.h"

namespace {

constexpr int NT = 256;

// tmp[p][y][ox] = sum_j src[p][y][xmin[ox] + j] * wx[ox][j]      (p = frame * 3 + channel)
__global__ __launch_bounds__(NT) void resize_h_kernel(const uint8_t* __restrict__ src, float* __restrict__ tmp,
                                                      const int* __restrict__ xmin, const int* __restrict__ xsize,
                                                      const float* __restrict__ wx, int taps, long rows, int W, int w) {
    const long total = rows * w;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int ox = (int)(i % w);
        const long row = i / w;
        const uint8_t* s = src + row * W + xmin[ox];
        const float* wt = wx + (long)ox * taps;
        const int n = xsize[ox];
        float t = (float)s[0] * wt[0];
        for (int j = 1; j < n; ++j) t += (float)s[j] * wt[j];
        tmp[i] = t;
    }
}

// dst[p][oy][ox] = clamp(rint(sum_j tmp[p][ymin[oy] + j][ox] * wy[oy][j]), 0, 255)
__global__ __launch_bounds__(NT) void resize_v_kernel(const float* __restrict__ tmp, uint8_t* __restrict__ dst,
                                                      const int* __restrict__ ymin, const int* __restrict__ ysize,
                                                      const float* __restrict__ wy, int taps, int planes, int H, int h, int w) {
    const long total = (long)planes * h * w;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int ox = (int)(i % w);
        const int oy = (int)((i / w) % h);
        const long p = i / ((long)w * h);
        const float* s = tmp + (p * H + ymin[oy]) * w + ox;
        const float* wt = wy + (long)oy * taps;
        const int n = ysize[oy];
        float t = s[0] * wt[0];
        for (int j = 1; j < n; ++j) t += s[(long)j * w] * wt[j];
        t = rintf(t);                                              // round half to even, as torch.round
        dst[i] = (uint8_t)fminf(fmaxf(t, 0.f), 255.f);
    }
}

// out[f] = src[idx[f]] for whole frames (uniform frame sampling: vision_process.py:252 linspace().round())
__global__ __launch_bounds__(NT) void gather_frames_kernel(const uint8_t* __restrict__ src, const int* __restrict__ idx,
                                                           uint8_t* __restrict__ out, int n, long frame_bytes) {
    const long per = frame_bytes >> 4;                             // 16-byte chunks per frame
    const long total = (long)n * per;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const long f = i / per, c = i % per;
        ((uint4*)(out + f * frame_bytes))[c] = ((const uint4*)(src + (long)idx[f] * frame_bytes))[c];
    }
}

inline int grid_for(long work) { long b = (work + NT - 1) / NT; return (int)(b < 1 ? 1 : (b > 256 * 32 ? 256 * 32 : b)); }

}  // namespace

extern "C" long spacer_resize_workspace_bytes(int planes, int H, int w) { return (long)planes * H * w * (long)sizeof(float); }

extern "C" int spacer_resize_bicubic_aa_u8(const uint8_t* src, uint8_t* dst, int planes, int H, int W, int h, int w,
                                           const int* xmin, const int* xsize, const float* wx, int taps_x, const int* ymin,
                                           const int* ysize, const float* wy, int taps_y, float* workspace,
                                           spacer_stream_t stream) {
    SP_REQUIRE(src && dst && xmin && xsize && wx && ymin && ysize && wy && workspace, SPACER_EINVAL, "resize: null operand");
    SP_REQUIRE(planes > 0 && H > 0 && W > 0 && h > 0 && w > 0 && taps_x > 0 && taps_y > 0, SPACER_EINVAL,
               "resize: bad shape planes=%d %dx%d -> %dx%d", planes, H, W, h, w);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(resize_h_kernel, dim3(grid_for((long)planes * H * w)), dim3(NT), 0, s, src, workspace, xmin, xsize, wx, taps_x,
                       (long)planes * H, W, w);
    hipLaunchKernelGGL(resize_v_kernel, dim3(grid_for((long)planes * h * w)), dim3(NT), 0, s, (const float*)workspace, dst, ymin, ysize,
                       wy, taps_y, planes, H, h, w);
    SP_CHECK_LAUNCH();
    return SPACER_OK;
}

extern "C" int spacer_gather_frames_u8(const uint8_t* src, const int* idx, uint8_t* out, int n, long frame_bytes,
                                       spacer_stream_t stream) {
    SP_REQUIRE(src && idx && out, SPACER_EINVAL, "gather_frames: null operand");
    SP_REQUIRE(frame_bytes % 16 == 0 && ((uintptr_t)src % 16) == 0 && ((uintptr_t)out % 16) == 0, SPACER_EINVAL,
               "gather_frames: frames must be 16-byte multiples and aligned (frame_bytes=%ld)", frame_bytes);
    if (n <= 0) return SPACER_OK;
    hipLaunchKernelGGL(gather_frames_kernel, dim3(grid_for((long)n * (frame_bytes >> 4))), dim3(NT), 0, (hipStream_t)stream, src, idx,
                       out, n, frame_bytes);
    SP_CHECK_LAUNCH();
    return SPACER_OK;
}
