// zr_post_oracle.cpp — TEST INFRASTRUCTURE (oracle).  CPU restatement of the reference's post stack, pinned against the
// genuine post_processor / bloom_filter by the fixtures of `zenith_ref post` (tests/golden/post_*.npz).  Follows
//   camera::process_framebuffer_to_image   camera.hpp:701-780
//   bloom_filter::generate_bloom_overlay   bloom.hpp:18-68
//   post_processor::process / apply_sharpening / analyze_framebuffer / apply_auto_exposure   color_processing.hpp:76-227
//   apply_aces, linear_to_gamma            common.hpp:48-84        vec3::luminance  vec3.hpp:106-108
// Only tests/ may call this; built into libzr_oracle.so with -ffp-contract=off.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../include/zr_capi.h"

namespace {

struct Col { double e[3]; };
inline Col operator*(const Col& c, double s) { return {{c.e[0] * s, c.e[1] * s, c.e[2] * s}}; }
inline Col operator+(const Col& a, const Col& b) { return {{a.e[0] + b.e[0], a.e[1] + b.e[1], a.e[2] + b.e[2]}}; }
inline Col operator-(const Col& a, const Col& b) { return {{a.e[0] - b.e[0], a.e[1] - b.e[1], a.e[2] - b.e[2]}}; }
inline Col over(const Col& c, double t) { return c * (1 / t); }   // vec3 operator/ multiplies by the reciprocal (vec3.hpp:77-79)
inline double lum(const Col& c) { return 0.2126 * c.e[0] + 0.7152 * c.e[1] + 0.0722 * c.e[2]; }
inline double gamma22(double v) { return v > 0 ? std::pow(v, 1.0 / 2.2) : 0.0; }

void blur(const std::vector<Col>& in, std::vector<Col>& out, int W, int H, int radius, bool horizontal) {
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            Col sum{{0, 0, 0}};
            float total = 0.0f;
            for (int off = -radius; off <= radius; ++off) {
                int sx = x + (horizontal ? off : 0), sy = y + (horizontal ? 0 : off);
                if (sx >= 0 && sx < W && sy >= 0 && sy < H) {
                    float w = 1.0f - (std::abs(off) / static_cast<float>(radius + 1));
                    sum = sum + in[(size_t)sy * W + sx] * static_cast<double>(w);
                    total += w;
                }
            }
            out[(size_t)y * W + x] = total > 0 ? over(sum, static_cast<double>(total)) : Col{{0, 0, 0}};
        }
}

Col rgb_to_hsv(const Col& c) {
    float r = (float)c.e[0], g = (float)c.e[1], b = (float)c.e[2];
    float mx = std::max({r, g, b}), mn = std::min({r, g, b});
    float h = 0.0f, s, v = mx, d = mx - mn;
    s = mx < 1e-6f ? 0.0f : d / mx;
    if (mx == mn) h = 0.0f;
    else {
        if (mx == r) h = (g - b) / d + (g < b ? 6.0f : 0.0f);
        else if (mx == g) h = (b - r) / d + 2.0f;
        else if (mx == b) h = (r - g) / d + 4.0f;
        h /= 6.0f;
    }
    return {{h * 360.0f, s, v}};
}

Col hsv_to_rgb(const Col& hsv) {
    float h = (float)hsv.e[0] / 360.0f, s = (float)hsv.e[1], v = (float)hsv.e[2];
    int i = (int)(h * 6.0f);
    float f = h * 6.0f - (float)i, p = v * (1.0f - s), q = v * (1.0f - f * s), t = v * (1.0f - (1.0f - f) * s);
    switch (i % 6) {
        case 0: return {{v, t, p}};
        case 1: return {{q, v, p}};
        case 2: return {{p, v, t}};
        case 3: return {{p, q, v}};
        case 4: return {{t, p, v}};
        case 5: return {{v, p, q}};
        default: return {{0, 0, 0}};
    }
}

double aces1(double v) {
    if (std::isnan(v) || std::isinf(v)) return 0.0;
    double val = std::max(0.0, v);
    const double a = 2.51, b = 0.03, c = 2.43, d = 0.59, e = 0.14;
    return (val * (a * val + b)) / (val * (c * val + d) + e);
}

Col process(Col exposed, float u, float v, const zr_post_params& pp) {
    Col c = exposed * static_cast<double>(pp.exposure);
    c = {{c.e[0] * pp.color_balance[0], c.e[1] * pp.color_balance[1], c.e[2] * pp.color_balance[2]}};
    if (std::abs(pp.contrast - 1.0f) > 0.001f) {
        double pivot = 0.18;
        for (double& x : c.e) x = std::max(0.0, (x - pivot) * pp.contrast + pivot);
    }
    if (pp.vignette_intensity > 0.0f) {
        float dist = std::sqrt((u - 0.5f) * (u - 0.5f) + (v - 0.5f) * (v - 0.5f));
        float vig = std::clamp(1.0f - dist * pp.vignette_intensity, 0.0f, 1.0f);
        c = c * static_cast<double>(vig);
    }
    if (std::abs(pp.saturation - 1.0f) > 0.001f || std::abs(pp.hue_shift) > 0.001f) {
        double luma = lum(c);
        if (luma > 0.0001) {
            Col hsv = rgb_to_hsv(over(c, luma));
            hsv.e[0] = std::fmod(hsv.e[0] + pp.hue_shift, 360.0f);
            if (hsv.e[0] < 0) hsv.e[0] += 360.0f;
            hsv.e[1] = std::clamp(static_cast<float>(hsv.e[1] * pp.saturation), 0.0f, 1.0f);
            c = hsv_to_rgb(hsv) * luma;
        }
    }
    if (pp.use_aces_tone_mapping) for (double& x : c.e) x = aces1(x);
    if (pp.debug_red || pp.debug_green || pp.debug_blue || pp.debug_luminance || pp.debug_bvh) {
        if (pp.debug_luminance) {
            double l = lum(c);
            if (l >= 1.0) c = {{1.0, 1.0, 1.0}};
            else if (l > 0.95) c = {{1.0, 0.0, 0.0}};
            else if (l > 0.70) c = {{1.0, 1.0, 0.0}};
            else if (l > 0.40) c = {{0.5, 0.5, 0.5}};
            else if (l > 0.10) c = {{0.0, 0.5, 0.0}};
            else if (l > 0.02) c = {{0.0, 0.0, 1.0}};
            else c = {{0.1, 0.0, 0.2}};
        } else if (!pp.debug_bvh) {
            c = {{pp.debug_red ? c.e[0] : 0.0, pp.debug_green ? c.e[1] : 0.0, pp.debug_blue ? c.e[2] : 0.0}};
        }
    }
    return {{gamma22(std::clamp(c.e[0], 0.0, 1.0)), gamma22(std::clamp(c.e[1], 0.0, 1.0)), gamma22(std::clamp(c.e[2], 0.0, 1.0))}};
}

}  // namespace

extern "C" {

int zro_post_process(const zr_post_params* pp, const double* frame, int W, int H, int is_data_pass, int apply_gamma, uint8_t* out) {
    const size_t n = (size_t)W * H;
    std::vector<Col> buffer(n);
    for (size_t i = 0; i < n; i++) buffer[i] = {{frame[3 * i], frame[3 * i + 1], frame[3 * i + 2]}};
    std::vector<Col> bloom_buffer = buffer;
    const double ev = std::pow(2.0, (double)pp->exposure);
    if (!is_data_pass && pp->use_bloom) {
        for (Col& p : bloom_buffer) p = p * ev;
        std::vector<Col> bright(n, Col{{0, 0, 0}}), tmp(n), overlay(n);
        for (size_t i = 0; i < n; ++i) {
            Col e = bloom_buffer[i] * static_cast<double>(1.0f);
            float l = static_cast<float>(lum(e));
            if (l > pp->bloom_threshold) {
                float factor = (l - pp->bloom_threshold) * pp->bloom_intensity;
                bright[i] = e * static_cast<double>(factor / std::max(l, 0.0001f));
            }
        }
        blur(bright, tmp, W, H, pp->bloom_radius, true);
        blur(tmp, overlay, W, H, pp->bloom_radius, false);
        const double inv_ev = 1.0 / ev;
        for (size_t i = 0; i < n; ++i) bloom_buffer[i] = buffer[i] + overlay[i] * inv_ev;
    }
    if (!is_data_pass && pp->use_sharpening && pp->sharpen_amount > 0.0) {
        std::vector<Col> orig = bloom_buffer;
        const double amount = pp->sharpen_amount;
        for (size_t y = 1; y < (size_t)H - 1; ++y)
            for (size_t x = 1; x < (size_t)W - 1; ++x) {
                size_t idx = y * W + x;
                Col sum = orig[idx] * 5.0;
                sum = sum - orig[(y - 1) * W + x];
                sum = sum - orig[(y + 1) * W + x];
                sum = sum - orig[y * W + (x - 1)];
                sum = sum - orig[y * W + (x + 1)];
                bloom_buffer[idx] = orig[idx] * (1.0 - amount) + sum * amount;
            }
    }
    for (int j = 0; j < H; j++)
        for (int i = 0; i < W; i++) {
            size_t k = (size_t)j * W + i;
            Col c = bloom_buffer[k];
            if (!is_data_pass) {
                c = c * ev;
                float u = static_cast<float>(i) / (W - 1), v = static_cast<float>(j) / (H - 1);
                c = process(c, u, v, *pp);
            } else {
                c = {{std::clamp(c.e[0], 0.0, 1.0), std::clamp(c.e[1], 0.0, 1.0), std::clamp(c.e[2], 0.0, 1.0)}};
                if (apply_gamma) c = {{gamma22(c.e[0]), gamma22(c.e[1]), gamma22(c.e[2])}};
            }
            out[3 * k] = static_cast<unsigned char>(255.999 * c.e[0]);
            out[3 * k + 1] = static_cast<unsigned char>(255.999 * c.e[1]);
            out[3 * k + 2] = static_cast<unsigned char>(255.999 * c.e[2]);
        }
    return ZR_OK;
}

int zro_analyze_frame(const double* frame, size_t n, zr_image_stats* st) {
    std::memset(st, 0, sizeof *st);
    double total = 0.0;
    for (size_t i = 0; i < n; i++) {
        float l = static_cast<float>(lum(Col{{frame[3 * i], frame[3 * i + 1], frame[3 * i + 2]}}));
        if (l > st->max_luminance) st->max_luminance = l;
        float cl = std::max(0.0001f, l);
        total += std::log2(cl);
        float ll = std::log2(cl);
        float nl = (ll - (-10.0f)) / 20.0f;
        int bin = std::clamp(static_cast<int>(nl * 255.0f), 0, 255);
        st->histogram[bin]++;
    }
    st->average_luminance = std::pow(2.0f, static_cast<float>(total / n));
    return ZR_OK;
}

// post_processor::apply_auto_exposure (color_processing.hpp:186-205)
double zro_auto_exposure(float average_luminance, float exposure, int use_auto_exposure, float target_luminance, float compensation_stops) {
    if (average_luminance <= 0.0) return static_cast<double>(exposure);
    if (!use_auto_exposure) return std::clamp(static_cast<float>(exposure), 0.01f, 10.0f);
    double safe = std::max(static_cast<double>(average_luminance), 0.02);
    double raw = target_luminance / safe;
    double cur = raw * std::pow(2.0, static_cast<double>(compensation_stops));
    return std::clamp(static_cast<float>(cur), 0.01f, 4.0f);
}

}  // extern "C"
