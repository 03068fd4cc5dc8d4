// clip_io.hip -- the data formats either side of the path: uint8 NHWC video frames in, uint8 NHWC
// HR frames out.
//
//   k_ingest_lr : what main.py:155-159 (MakeDataDatasetToTensor) does per clip item with three ATen ops and two
//                 transposes -- `interpolate(transpose1323(d.float()), (int(H/4), int(W/4)))` back to NHWC -- as one pass:
//                 uint8 [F,H,W,3] -> float32 LR [F,h,w,3], nearest neighbour with ATen's index rule
//                 src = min(floor(dst * (float)in / out), in - 1).  Reads 3 of every 3*H/h bytes of every (H/h)-th row.
//   k_u8_to_f32 : `datas.type(torch.float32)` (main.py:161-167: target / high_frames), 4 bytes in -> 16 bytes out per thread.
//   k_f32_to_u8 : HR write-out: round to nearest (ties to even, like numpy's rint) and clamp to 0..255; 16 B in -> 4 B out.
// All three are HBM-bound byte passes: one thread = one pixel (or four values), lanes along the fastest axis.
#include "vsr_common.h"

namespace {

__global__ void __launch_bounds__(256)
k_ingest_lr(const unsigned char* __restrict__ in, float* __restrict__ lr, int H, int W, int h, int w, float sy, float sx) {
    const int f = blockIdx.z, y = blockIdx.y;
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= w) return;
    int yy = (int)floorf((float)y * sy), xx = (int)floorf((float)x * sx);   // ATen nearest_neighbor_compute_source_index
    yy = yy < H - 1 ? yy : H - 1;
    xx = xx < W - 1 ? xx : W - 1;
    const unsigned char* p = in + (((size_t)f * H + yy) * W + xx) * 3;
    float* o = lr + (((size_t)f * h + y) * w + x) * 3;
    o[0] = (float)p[0];
    o[1] = (float)p[1];
    o[2] = (float)p[2];
}

__global__ void __launch_bounds__(256) k_u8_to_f32(const unsigned char* __restrict__ in, float* __restrict__ out, size_t n) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 4 <= n) {
        const uchar4 v = *reinterpret_cast<const uchar4*>(in + i);
        *reinterpret_cast<float4*>(out + i) = make_float4((float)v.x, (float)v.y, (float)v.z, (float)v.w);
    } else {
        for (size_t j = i; j < n; ++j) out[j] = (float)in[j];
    }
}

__global__ void __launch_bounds__(256) k_f32_to_u8(const float* __restrict__ in, unsigned char* __restrict__ out, size_t n) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    auto q = [](float v) -> unsigned char {
        v = rintf(v);                    // round half to even
        if (!(v >= 0.0f)) v = 0.0f;      // negatives and NaN
        if (v > 255.0f) v = 255.0f;
        return (unsigned char)v;
    };
    if (i + 4 <= n) {
        const float4 v = *reinterpret_cast<const float4*>(in + i);
        *reinterpret_cast<uchar4*>(out + i) = make_uchar4(q(v.x), q(v.y), q(v.z), q(v.w));
    } else {
        for (size_t j = i; j < n; ++j) out[j] = q(in[j]);
    }
}

}  // namespace

extern "C" {

int vsr_clip_ingest_u8(const void* frames_u8, float* lr, float* hr_or_null, int F, int H, int W, int h, int w,
                       vsr_stream_t stream) {
    VSR_REQUIRE(frames_u8 && lr, "clip_ingest: null pointer");
    VSR_REQUIRE(F > 0 && H > 0 && W > 0 && h > 0 && w > 0 && h <= H && w <= W && h <= 65535 && F <= 65535, "clip_ingest: bad shape");
    VSR_REQUIRE((reinterpret_cast<uintptr_t>(frames_u8) & 3) == 0 && (!hr_or_null || (reinterpret_cast<uintptr_t>(hr_or_null) & 15) == 0),
                "clip_ingest: frames must be 4-byte aligned, the float copy 16-byte aligned");
    // scale as ATen forms it when only the output size is given (compute_scales_value): (float)input / output
    const float sy = (float)H / (float)h, sx = (float)W / (float)w;
    hipLaunchKernelGGL(k_ingest_lr, dim3(vsr::cdiv(w, 256), h, F), dim3(256), 0, vsr::S(stream), (const unsigned char*)frames_u8, lr,
                       H, W, h, w, sy, sx);
    int rc = vsr::launched("clip_ingest/lr");
    if (rc || !hr_or_null) return rc;
    const size_t n = (size_t)F * H * W * 3;
    hipLaunchKernelGGL(k_u8_to_f32, dim3(vsr::cdiv((long long)((n + 3) / 4), 256)), dim3(256), 0, vsr::S(stream),
                       (const unsigned char*)frames_u8, hr_or_null, n);
    return vsr::launched("clip_ingest/hr");
}

int vsr_frame_to_u8(const float* frame, void* out_u8, size_t n, vsr_stream_t stream) {
    VSR_REQUIRE(frame && out_u8 && n > 0, "frame_to_u8: null pointer / empty frame");
    VSR_REQUIRE((reinterpret_cast<uintptr_t>(frame) & 15) == 0 && (reinterpret_cast<uintptr_t>(out_u8) & 3) == 0, "frame_to_u8: alignment");
    hipLaunchKernelGGL(k_f32_to_u8, dim3(vsr::cdiv((long long)((n + 3) / 4), 256)), dim3(256), 0, vsr::S(stream), frame,
                       (unsigned char*)out_u8, n);
    return vsr::launched("frame_to_u8");
}

}  // extern "C"
