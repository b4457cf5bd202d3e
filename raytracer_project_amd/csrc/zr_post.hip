// zr_post.hip — the steps after the sample loop on the device: camera::process_framebuffer_to_image
// (camera.hpp:701-780) = bloom (bloom.hpp:18-68) -> sharpen (color_processing.hpp:207-227) -> exposure + post_processor::process
// (color_processing.hpp:76-147: colour balance, contrast, vignette, HSV, ACES, debug views, clamp, gamma) -> 8-bit RGB, and
// post_processor::analyze_framebuffer (color_processing.hpp:150-183: log-luminance histogram and mean for auto-exposure).
//
// One thread per pixel; every kernel is a single pass over W*H double3 (HBM-bound, a few MB per frame).  The arithmetic keeps
// the reference's mix of float and double and its operation order; this file is compiled with -ffp-contract=off so that no
// multiply-add is fused and the 8-bit result is the reference's byte for byte.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/zr_capi.h"

namespace zr {

namespace {

struct C3 { double x, y, z; };
__device__ __forceinline__ C3 ld(const double* p, size_t i) { C3 c; c.x = p[3 * i]; c.y = p[3 * i + 1]; c.z = p[3 * i + 2]; return c; }
__device__ __forceinline__ void st(double* p, size_t i, C3 c) { p[3 * i] = c.x; p[3 * i + 1] = c.y; p[3 * i + 2] = c.z; }
__device__ __forceinline__ C3 mul(C3 c, double s) { C3 r; r.x = c.x * s; r.y = c.y * s; r.z = c.z * s; return r; }
__device__ __forceinline__ C3 add(C3 a, C3 b) { C3 r; r.x = a.x + b.x; r.y = a.y + b.y; r.z = a.z + b.z; return r; }
__device__ __forceinline__ C3 sub(C3 a, C3 b) { C3 r; r.x = a.x - b.x; r.y = a.y - b.y; r.z = a.z - b.z; return r; }
__device__ __forceinline__ double luminance(C3 c) { return 0.2126 * c.x + 0.7152 * c.y + 0.0722 * c.z; }  // vec3.hpp:106-108
__device__ __forceinline__ double clamp01(double v) { return v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v); }       // std::clamp(v, 0.0, 1.0)
__device__ __forceinline__ double gamma22(double v) { return v > 0 ? pow(v, 1.0 / 2.2) : 0.0; }           // common.hpp:70-75

// bright pass of bloom_filter::generate_bloom_overlay with exposure = 1.0f on the EV-scaled buffer (camera.hpp:714-722, bloom.hpp:28-39)
__global__ void post_bright(const double* __restrict__ frame, size_t n, double ev, float threshold, float intensity, double* __restrict__ bright) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    C3 e = mul(mul(ld(frame, i), ev), (double)1.0f);
    const float lum = (float)luminance(e);
    C3 o; o.x = o.y = o.z = 0.0;
    if (lum > threshold) {
        const float factor = (lum - threshold) * intensity;
        o = mul(e, (double)(factor / fmaxf(lum, 0.0001f)));
    }
    st(bright, i, o);
}

// bloom_filter::blur_pass (bloom.hpp:45-67): linear-falloff kernel, offsets in increasing order
__global__ void post_blur(const double* __restrict__ in, double* __restrict__ out, int W, int H, int radius, int horizontal) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)W * H) return;
    const int x = (int)(i % W), y = (int)(i / W);
    C3 sum; sum.x = sum.y = sum.z = 0.0;
    float total = 0.0f;
    for (int off = -radius; off <= radius; ++off) {
        const int sx = x + (horizontal ? off : 0), sy = y + (horizontal ? 0 : off);
        if (sx >= 0 && sx < W && sy >= 0 && sy < H) {
            const float wgt = 1.0f - ((float)abs(off) / (float)(radius + 1));
            sum = add(sum, mul(ld(in, (size_t)sy * W + sx), (double)wgt));
            total += wgt;
        }
    }
    C3 o; o.x = o.y = o.z = 0.0;
    if (total > 0) {   // vec3 operator/ is a multiplication by the reciprocal (vec3.hpp:77-79)
        const double inv = 1 / (double)total;
        o = mul(sum, inv);
    }
    st(out, i, o);
}

// bloom_buffer[i] = buffer[i] + overlay[i] * inv_ev (camera.hpp:724-727)
__global__ void post_bloom_add(const double* __restrict__ frame, const double* __restrict__ overlay, size_t n, double inv_ev, double* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    st(out, i, add(ld(frame, i), mul(ld(overlay, i), inv_ev)));
}

// post_processor::apply_sharpening (color_processing.hpp:207-227): interior pixels only
__global__ void post_sharpen(const double* __restrict__ in, double* __restrict__ out, int W, int H, double amount) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)W * H) return;
    const int x = (int)(i % W), y = (int)(i / W);
    C3 c = ld(in, i);
    if (x >= 1 && x < W - 1 && y >= 1 && y < H - 1) {
        C3 sum = mul(c, 5.0);
        sum = sub(sum, ld(in, i - W));
        sum = sub(sum, ld(in, i + W));
        sum = sub(sum, ld(in, i - 1));
        sum = sub(sum, ld(in, i + 1));
        c = add(mul(c, 1.0 - amount), mul(sum, amount));
    }
    st(out, i, c);
}

__device__ inline C3 rgb_to_hsv(C3 c) {  // color_processing.hpp:286-312 (float arithmetic)
    const float r = (float)c.x, g = (float)c.y, b = (float)c.z;
    const float mx = fmaxf(fmaxf(r, g), b), mn = fminf(fminf(r, g), b);
    float h = 0.0f;
    const float v = mx, d = mx - mn;
    const float s = mx < 1e-6f ? 0.0f : d / mx;
    if (mx == mn) h = 0.0f;
    else {
        if (mx == r) h = (g - b) / d + (g < b ? 6.0f : 0.0f);
        else if (mx == g) h = (b - r) / d + 2.0f;
        else if (mx == b) h = (r - g) / d + 4.0f;
        h /= 6.0f;
    }
    C3 o; o.x = h * 360.0f; o.y = s; o.z = v;
    return o;
}

__device__ inline C3 hsv_to_rgb(C3 hsv) {  // color_processing.hpp:314-344
    const float h = (float)hsv.x / 360.0f, s = (float)hsv.y, v = (float)hsv.z;
    const int i = (int)(h * 6.0f);
    const float f = h * 6.0f - (float)i;
    const float p = v * (1.0f - s), q = v * (1.0f - f * s), t = v * (1.0f - (1.0f - f) * s);
    C3 o;
    switch (i % 6) {
        case 0: o.x = v; o.y = t; o.z = p; break;
        case 1: o.x = q; o.y = v; o.z = p; break;
        case 2: o.x = p; o.y = v; o.z = t; break;
        case 3: o.x = p; o.y = q; o.z = v; break;
        case 4: o.x = t; o.y = p; o.z = v; break;
        case 5: o.x = v; o.y = p; o.z = q; break;
        default: o.x = o.y = o.z = 0.0; break;
    }
    return o;
}

__device__ inline double aces1(double v) {  // common.hpp:48-67
    if (isnan(v) || isinf(v)) return 0.0;
    const double val = fmax(0.0, v);
    const double a = 2.51, b = 0.03, c = 2.43, d = 0.59, e = 0.14;
    return (val * (a * val + b)) / (val * (c * val + d) + e);
}

// post_processor::process for the beauty pass (color_processing.hpp:76-147)
__device__ inline C3 process_rgb(C3 exposed, float u, float v, const zr_post_params& pp) {
    C3 c = mul(exposed, (double)pp.exposure);
    c.x = c.x * pp.color_balance[0]; c.y = c.y * pp.color_balance[1]; c.z = c.z * pp.color_balance[2];
    if (fabsf(pp.contrast - 1.0f) > 0.001f) {   // apply_contrast, pivot 0.18
        const double pivot = 0.18;
        c.x = fmax(0.0, (c.x - pivot) * pp.contrast + pivot);
        c.y = fmax(0.0, (c.y - pivot) * pp.contrast + pivot);
        c.z = fmax(0.0, (c.z - pivot) * pp.contrast + pivot);
    }
    if (pp.vignette_intensity > 0.0f) {
        const float dist = sqrtf((u - 0.5f) * (u - 0.5f) + (v - 0.5f) * (v - 0.5f));
        const float t = 1.0f - dist * pp.vignette_intensity;
        const float vig = t < 0.0f ? 0.0f : (t > 1.0f ? 1.0f : t);
        c = mul(c, (double)vig);
    }
    if (fabsf(pp.saturation - 1.0f) > 0.001f || fabsf(pp.hue_shift) > 0.001f) {
        const double luma = luminance(c);
        if (luma > 0.0001) {
            C3 hsv = rgb_to_hsv(mul(c, 1 / luma));
            hsv.x = fmod(hsv.x + pp.hue_shift, (double)360.0f);
            if (hsv.x < 0) hsv.x += 360.0f;
            const float s = (float)(hsv.y * pp.saturation);
            hsv.y = s < 0.0f ? 0.0f : (s > 1.0f ? 1.0f : s);
            c = mul(hsv_to_rgb(hsv), luma);
        }
    }
    if (pp.use_aces_tone_mapping) { c.x = aces1(c.x); c.y = aces1(c.y); c.z = aces1(c.z); }
    if (pp.debug_red || pp.debug_green || pp.debug_blue || pp.debug_luminance || pp.debug_bvh) {   // apply_debug_view
        if (pp.debug_luminance) {
            const double lum = luminance(c);
            if (lum >= 1.0) { c.x = 1.0; c.y = 1.0; c.z = 1.0; }
            else if (lum > 0.95) { c.x = 1.0; c.y = 0.0; c.z = 0.0; }
            else if (lum > 0.70) { c.x = 1.0; c.y = 1.0; c.z = 0.0; }
            else if (lum > 0.40) { c.x = 0.5; c.y = 0.5; c.z = 0.5; }
            else if (lum > 0.10) { c.x = 0.0; c.y = 0.5; c.z = 0.0; }
            else if (lum > 0.02) { c.x = 0.0; c.y = 0.0; c.z = 1.0; }
            else { c.x = 0.1; c.y = 0.0; c.z = 0.2; }
        } else if (!pp.debug_bvh) {
            c.x = pp.debug_red ? c.x : 0.0; c.y = pp.debug_green ? c.y : 0.0; c.z = pp.debug_blue ? c.z : 0.0;
        }
    }
    C3 o; o.x = gamma22(clamp01(c.x)); o.y = gamma22(clamp01(c.y)); o.z = gamma22(clamp01(c.z));
    return o;
}

// tone mapping and RGB conversion (camera.hpp:741-773)
__global__ void post_final(const double* __restrict__ in, int W, int H, double ev, zr_post_params pp, int is_data_pass, int apply_gamma,
                           uint8_t* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)W * H) return;
    C3 c = ld(in, i);
    if (!is_data_pass) {
        c = mul(c, ev);
        const float u = (float)(int)(i % W) / (float)(W - 1), v = (float)(int)(i / W) / (float)(H - 1);
        c = process_rgb(c, u, v, pp);
    } else {
        c.x = clamp01(c.x); c.y = clamp01(c.y); c.z = clamp01(c.z);
        if (apply_gamma) { c.x = gamma22(c.x); c.y = gamma22(c.y); c.z = gamma22(c.z); }
    }
    out[3 * i] = (unsigned char)(255.999 * c.x);
    out[3 * i + 1] = (unsigned char)(255.999 * c.y);
    out[3 * i + 2] = (unsigned char)(255.999 * c.z);
}

// analyze_framebuffer (color_processing.hpp:150-183): per-block partial sums; the host adds the partials in block order
__global__ __launch_bounds__(256) void post_analyze(const double* __restrict__ frame, size_t n, double* __restrict__ part_log, float* __restrict__ part_max,
                                                     int* __restrict__ hist) {
    __shared__ double s_log[256];
    __shared__ float s_max[256];
    __shared__ int s_hist[256];
    s_hist[threadIdx.x] = 0;
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    double lg = 0.0; float mx = 0.0f;
    if (i < n) {
        const float lum = (float)luminance(ld(frame, i));
        mx = lum;
        const float cl = fmaxf(0.0001f, lum);
        const float l2 = log2f(cl);
        lg = (double)l2;
        const float nl = (l2 - (-10.0f)) / 20.0f;
        int bin = (int)(nl * 255.0f);
        bin = bin < 0 ? 0 : (bin > 255 ? 255 : bin);
        atomicAdd(&s_hist[bin], 1);
    }
    s_log[threadIdx.x] = lg; s_max[threadIdx.x] = mx;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) { s_log[threadIdx.x] += s_log[threadIdx.x + s]; s_max[threadIdx.x] = fmaxf(s_max[threadIdx.x], s_max[threadIdx.x + s]); }
        __syncthreads();
    }
    if (threadIdx.x == 0) { part_log[blockIdx.x] = s_log[0]; part_max[blockIdx.x] = s_max[0]; }
    if (s_hist[threadIdx.x]) atomicAdd(&hist[threadIdx.x], s_hist[threadIdx.x]);
}

inline unsigned grid_for(size_t n) { return (unsigned)((n + 255) / 256); }

}  // namespace

// d_frame: W*H*3 doubles; d_tmp0 / d_tmp1 / d_tmp2: scratch of the same size; d_out: W*H*3 bytes
hipError_t launch_post(const double* d_frame, int W, int H, const zr_post_params& pp, int is_data_pass, int apply_gamma, double ev, double* d_tmp0,
                       double* d_tmp1, double* d_tmp2, uint8_t* d_out, hipStream_t stream) {
    const size_t n = (size_t)W * H;
    const double* cur = d_frame;
    if (!is_data_pass && pp.use_bloom) {
        hipLaunchKernelGGL(post_bright, dim3(grid_for(n)), dim3(256), 0, stream, d_frame, n, ev, pp.bloom_threshold, pp.bloom_intensity, d_tmp0);
        hipLaunchKernelGGL(post_blur, dim3(grid_for(n)), dim3(256), 0, stream, d_tmp0, d_tmp1, W, H, pp.bloom_radius, 1);
        hipLaunchKernelGGL(post_blur, dim3(grid_for(n)), dim3(256), 0, stream, d_tmp1, d_tmp0, W, H, pp.bloom_radius, 0);
        hipLaunchKernelGGL(post_bloom_add, dim3(grid_for(n)), dim3(256), 0, stream, d_frame, d_tmp0, n, 1.0 / ev, d_tmp1);
        cur = d_tmp1;
    }
    if (!is_data_pass && pp.use_sharpening && pp.sharpen_amount > 0.0) {
        hipLaunchKernelGGL(post_sharpen, dim3(grid_for(n)), dim3(256), 0, stream, cur, d_tmp2, W, H, pp.sharpen_amount);
        cur = d_tmp2;
    }
    hipLaunchKernelGGL(post_final, dim3(grid_for(n)), dim3(256), 0, stream, cur, W, H, ev, pp, is_data_pass, apply_gamma, d_out);
    return hipGetLastError();
}

hipError_t launch_analyze(const double* d_frame, size_t n, double* d_part_log, float* d_part_max, int* d_hist, hipStream_t stream) {
    hipError_t e = hipMemsetAsync(d_hist, 0, 256 * sizeof(int), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(post_analyze, dim3(grid_for(n)), dim3(256), 0, stream, d_frame, n, d_part_log, d_part_max, d_hist);
    return hipGetLastError();
}

// ---- multi-GPU exchange (zr_comm.cpp): frame pixels <-> a rank's packed tile list; pure data movement -----------------------
__global__ __launch_bounds__(256) void pack_tiles(const double* __restrict__ frame, const uint32_t* __restrict__ idx, size_t n, double* __restrict__ packed) {
    const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const double* p = frame + (size_t)idx[k] * 3;
    packed[k * 3] = p[0]; packed[k * 3 + 1] = p[1]; packed[k * 3 + 2] = p[2];
}
__global__ __launch_bounds__(256) void unpack_tiles(double* __restrict__ frame, const uint32_t* __restrict__ idx, size_t n, const double* __restrict__ packed) {
    const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    double* p = frame + (size_t)idx[k] * 3;
    p[0] = packed[k * 3]; p[1] = packed[k * 3 + 1]; p[2] = packed[k * 3 + 2];
}
hipError_t launch_pack_tiles(const double* frame, const uint32_t* idx, size_t n, double* packed, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(pack_tiles, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, frame, idx, n, packed);
    return hipGetLastError();
}
hipError_t launch_unpack_tiles(double* frame, const uint32_t* idx, size_t n, const double* packed, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(unpack_tiles, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, frame, idx, n, packed);
    return hipGetLastError();
}

}  // namespace zr
