// Per-ray kernels of the HyperNeRF render path for gfx950: stratified sampling, density activation +
// alpha compositing (forward/backward), inverse-CDF hierarchical sampling with merge sort, GLO
// embedding gather / gradient.  One wavefront (64 lanes) per ray; all HBM accesses are coalesced
// rows of the (B,S,*) arrays.  These kernels are HBM-bound (tens of bytes per sample).
#include "hn_common.h"
#include "hn_pack.h"

// ------------------------------------------------------------------------------------------------
// wave64 scans — on the DPP cross-lane paths of CDNA (round 6; rounds 1-5 went through `__shfl_up` = `ds_bpermute_b32`, an
// LDS-pipe round trip per step): an inclusive scan is Kogge-Stone inside each row of 16 lanes (row_shr:1, 2, 4, 8; a lane
// without a source keeps the identity), then lane 15 of rows 0 / 2 joins rows 1 / 3 (row_bcast:15, row mask 0xa) and lane
// 31 rows 2-3 (row_bcast:31, row mask 0xc): six `v_mov_b32_dpp` + six ALU ops, no LDS traffic, no wait.
// ------------------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK>
HN_DEV float hn_dpp(float old, float src) {      // lanes whose source lane does not exist (or whose row is masked) return `old`
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src),
                                                               CTRL, ROW_MASK, 0xf, false));
}
constexpr int HN_DPP_ROW_SHR = 0x110, HN_DPP_WAVE_SHR1 = 0x138, HN_DPP_BCAST15 = 0x142, HN_DPP_BCAST31 = 0x143;
HN_DEV float hn_wave_incl_scan_mul(float v, int) {
  v *= hn_dpp<HN_DPP_ROW_SHR + 1, 0xf>(1.0f, v);
  v *= hn_dpp<HN_DPP_ROW_SHR + 2, 0xf>(1.0f, v);
  v *= hn_dpp<HN_DPP_ROW_SHR + 4, 0xf>(1.0f, v);
  v *= hn_dpp<HN_DPP_ROW_SHR + 8, 0xf>(1.0f, v);
  v *= hn_dpp<HN_DPP_BCAST15, 0xa>(1.0f, v);
  v *= hn_dpp<HN_DPP_BCAST31, 0xc>(1.0f, v);
  return v;
}
HN_DEV float hn_wave_incl_scan_add(float v, int) {
  v += hn_dpp<HN_DPP_ROW_SHR + 1, 0xf>(0.0f, v);
  v += hn_dpp<HN_DPP_ROW_SHR + 2, 0xf>(0.0f, v);
  v += hn_dpp<HN_DPP_ROW_SHR + 4, 0xf>(0.0f, v);
  v += hn_dpp<HN_DPP_ROW_SHR + 8, 0xf>(0.0f, v);
  v += hn_dpp<HN_DPP_BCAST15, 0xa>(0.0f, v);
  v += hn_dpp<HN_DPP_BCAST31, 0xc>(0.0f, v);
  return v;
}
// the value of the lane below (identity in lane 0): wave_shr:1
HN_DEV float hn_wave_shr1(float v, float identity) { return hn_dpp<HN_DPP_WAVE_SHR1, 0xf>(identity, v); }
// one lane's value in every lane (`lane_idx` wave-uniform): v_readlane_b32, a scalar-register broadcast
HN_DEV float hn_wave_bcast(float v, int lane_idx) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane_idx));
}
HN_DEV float hn_wave_sum(float v) { return hn_wave_bcast(hn_wave_incl_scan_add(v, 0), 63); }

// ------------------------------------------------------------------------------------------------
// sampling along rays
// ------------------------------------------------------------------------------------------------
__global__ void hn_sample_kernel(const float* __restrict__ origins, const float* __restrict__ dirs, int ray_ld,
                                 const float* __restrict__ lower, const float* __restrict__ upper,
                                 int per_ray_bounds, const float* __restrict__ t_rand, float scale, int n_rays,
                                 int n, float* __restrict__ z_out, float* __restrict__ pts_out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)n_rays * n) return;
  const int b = (int)(i / n), s = (int)(i % n);
  const size_t bi = per_ray_bounds ? i : (size_t)s;
  float z = lower[bi];
  if (t_rand != nullptr) {
    // z = lower + (upper - lower) * (scale * t): three separately rounded fp32 ops, as ATen does
    float t = t_rand[i];
    if (scale != 1.0f) t = __fmul_rn(scale, t);
    z = __fadd_rn(z, __fmul_rn(__fsub_rn(upper[bi], z), t));
  }
  z_out[i] = z;
  if (pts_out != nullptr) {
#pragma unroll
    for (int c = 0; c < 3; ++c)
      pts_out[i * 3 + c] = __fadd_rn(origins[(size_t)b * ray_ld + c], __fmul_rn(z, dirs[(size_t)b * ray_ld + c]));
  }
}

extern "C" int hn_sample_along_rays(const float* origins, const float* dirs, int ray_ld, const float* lower,
                                    const float* upper, int per_ray_bounds, const float* t_rand, float scale,
                                    int n_rays, int n, float* z_out, float* pts_out, hnStream_t stream) {
  if (n_rays <= 0 || n <= 0) return -2;
  if (lower == nullptr || z_out == nullptr) return -3;
  if (t_rand != nullptr && upper == nullptr) return -3;
  if (pts_out != nullptr && (origins == nullptr || dirs == nullptr)) return -3;
  const size_t total = (size_t)n_rays * n;
  hipLaunchKernelGGL(hn_sample_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     origins, dirs, ray_ld, lower, upper, per_ray_bounds, t_rand, scale, n_rays, n, z_out, pts_out);
  HN_CHECK_LAUNCH();
  return 0;
}

// legacy nerf_pl sampling: per-ray near/far (models/rendering.py:189-207)
__global__ void hn_sample_legacy_kernel(const float* __restrict__ rays, int ray_ld, const float* __restrict__ t_vals,
                                        const float* __restrict__ omt_vals, int use_disp,
                                        const float* __restrict__ t_rand, float scale, int n_rays, int n,
                                        float* __restrict__ z_out, float* __restrict__ pts_out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)n_rays * n) return;
  const int b = (int)(i / n), s = (int)(i % n);
  const float* ray = rays + (size_t)b * ray_ld;
  const float near = ray[6], far = ray[7];
  auto zlin = [&](int k) -> float {
    if (!use_disp) return __fadd_rn(__fmul_rn(near, omt_vals[k]), __fmul_rn(far, t_vals[k]));
    const float den = __fadd_rn(__fmul_rn(__fdiv_rn(1.0f, near), omt_vals[k]), __fmul_rn(__fdiv_rn(1.0f, far), t_vals[k]));
    return __fdiv_rn(1.0f, den);
  };
  float z = zlin(s);
  if (t_rand != nullptr) {
    const float lower = s == 0 ? z : __fmul_rn(0.5f, __fadd_rn(zlin(s - 1), z));
    const float upper = s == n - 1 ? z : __fmul_rn(0.5f, __fadd_rn(z, zlin(s + 1)));
    const float t = __fmul_rn(scale, t_rand[i]);
    z = __fadd_rn(lower, __fmul_rn(__fsub_rn(upper, lower), t));
  }
  z_out[i] = z;
  if (pts_out != nullptr) {
#pragma unroll
    for (int c = 0; c < 3; ++c) pts_out[i * 3 + c] = __fadd_rn(ray[c], __fmul_rn(ray[3 + c], z));
  }
}

extern "C" int hn_sample_legacy(const float* rays, int ray_ld, const float* t_vals, const float* one_minus_t,
                                int use_disp, const float* t_rand, float scale, int n_rays, int n, float* z_out,
                                float* pts_out, hnStream_t stream) {
  if (n_rays <= 0 || n <= 0 || ray_ld < 8) return -2;
  if (rays == nullptr || t_vals == nullptr || one_minus_t == nullptr || z_out == nullptr) return -3;
  const size_t total = (size_t)n_rays * n;
  hipLaunchKernelGGL(hn_sample_legacy_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     rays, ray_ld, t_vals, one_minus_t, use_disp, t_rand, scale, n_rays, n, z_out, pts_out);
  HN_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// stand-alone positional encoders (the fused path generates these features inside the MLP machine)
// ------------------------------------------------------------------------------------------------
__global__ void hn_posenc_kernel(const float* __restrict__ x, size_t n, int c, const float* __restrict__ freqs,
                                 int n_freqs, int identity, int jax_cos, float* __restrict__ out,
                                 const float* __restrict__ g_out, float* __restrict__ g_x) {
  const int width = c * (2 * n_freqs + (identity ? 1 : 0));
  if (g_out == nullptr) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * (size_t)width) return;
    const size_t row = i / width;
    int f = (int)(i % width);
    float v;
    if (identity && f < c) {
      v = x[row * c + f];
    } else {
      if (identity) f -= c;
      const int k = f / (2 * c), rem = f % (2 * c), sc = rem / c, ch = rem % c;
      float arg = __fmul_rn(freqs[k], x[row * c + ch]);
      if (sc == 0) v = sinf(arg);
      else v = jax_cos ? sinf(__fadd_rn(arg, 0.5f * 3.1415926f)) : cosf(arg);
    }
    out[i] = v;
  } else {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * (size_t)c) return;
    const size_t row = i / c;
    const int ch = (int)(i % c);
    const float xv = x[i];
    const float* g = g_out + row * width;
    float acc = identity ? g[ch] : 0.0f;
    const int base = identity ? c : 0;
    for (int k = 0; k < n_freqs; ++k) {
      const float fr = freqs[k];
      const float arg = __fmul_rn(fr, xv);
      acc += g[base + k * 2 * c + ch] * fr * cosf(arg);
      if (jax_cos) acc += g[base + k * 2 * c + c + ch] * fr * cosf(__fadd_rn(arg, 0.5f * 3.1415926f));
      else acc -= g[base + k * 2 * c + c + ch] * fr * sinf(arg);
    }
    g_x[i] = acc;
  }
}

extern "C" int hn_posenc(const float* x, int64_t n, int c, const float* freqs, int n_freqs, int identity, int jax_cos,
                         float* out, const float* g_out, float* g_x, hnStream_t stream) {
  if (n <= 0 || c <= 0 || n_freqs < 0) return -2;
  if (x == nullptr || (n_freqs > 0 && freqs == nullptr)) return -3;
  if ((g_out == nullptr) == (out == nullptr)) return -3;   // exactly one of forward / backward
  if (g_out != nullptr && g_x == nullptr) return -3;
  const int width = c * (2 * n_freqs + (identity ? 1 : 0));
  const size_t total = g_out == nullptr ? (size_t)n * width : (size_t)n * c;
  hipLaunchKernelGGL(hn_posenc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                     (size_t)n, c, freqs, n_freqs, identity, jax_cos, out, g_out, g_x);
  HN_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// compositing
// ------------------------------------------------------------------------------------------------
constexpr int HN_MAX_SEG = 8;  // up to 512 samples per ray

HN_DEV float hn_softplus(float x) { return x > 20.0f ? x : log1pf(expf(x)); }  // torch Softplus(beta=1,threshold=20)
HN_DEV float hn_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// Where sorted sample s of a ray lives when the level was evaluated in two parts (HnCompositeArgs.perm): the part
// (0 = the coarse level's samples, 1 = the new ones) and the row inside that part's (B, n, .) arrays.
struct HnPartRow {
  size_t row;
  bool second;
};
HN_DEV HnPartRow hn_part_row(const HnCompositeArgs& a, int ray, size_t row, int s, const int* perm_lds = nullptr) {
  if (a.perm == nullptr) return HnPartRow{row + s, false};
  const int k = perm_lds != nullptr ? perm_lds[s] : a.perm[row + s];
  if (k < a.split) return HnPartRow{(size_t)ray * a.split + k, false};
  return HnPartRow{(size_t)ray * (a.n_samples - a.split) + (k - a.split), true};
}

// One ray's compositing on one wave.  `w_lds` (forward only, optional): the ray's weights are ALSO left in LDS there — the
// inverse-CDF sampler of the same wave reads them from it (hn_composite_pdf_kernel).
template <bool BACKWARD>
HN_DEV void hn_composite_ray(const HnCompositeArgs& a, int ray, int lane, float* w_lds, int* perm_lds = nullptr) {
  const int S = a.n_samples;
  const int nseg = (S + 63) / 64;
  const size_t row = (size_t)ray * S;
  const float* d = a.dirs + (size_t)ray * a.ray_ld;
  const float dnorm = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  const float eps = a.variant != 1 ? 1e-5f : 1e-10f;
  const float last = a.variant != 1 ? (a.sample_at_infinity ? 1e7f : 1e-7f) : 1e10f;

  // a level in two parts: the merge permutation is staged in LDS once (round 6) — a sample's rgb / density address then
  // depends on no other global load (perm, then the row: two memory latencies per segment and direction before)
  if (a.perm == nullptr) perm_lds = nullptr;
  if (perm_lds != nullptr) {
    for (int i = lane; i < S; i += 64) perm_lds[i] = a.perm[row + i];
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
  }
  float alpha[HN_MAX_SEG], om[HN_MAX_SEG], trans[HN_MAX_SEG], wgt[HN_MAX_SEG], pre[HN_MAX_SEG], dist[HN_MAX_SEG];
  float carry = 1.0f;   // product of (1-alpha+eps) over all earlier segments
  float csum = 0.0f;    // running sum of weights (median depth)
  float s_r = 0.f, s_g = 0.f, s_b = 0.f, s_d = 0.f, s_w = 0.f, s_wl = 0.f;
  float med_z = 0.0f, med_pt = 0.0f;
  bool med_found = false;
#pragma unroll
  for (int k = 0; k < HN_MAX_SEG; ++k) {
    if (k < nseg) {
      const int s = k * 64 + lane;
      const bool in = s < S;
      float zz = 0.f, zn = 0.f, raw = 0.f, cr = 0.f, cg = 0.f, cb = 0.f;
      HnPartRow pr{row, false};
      if (in) {
        zz = a.z[row + s];
        zn = (s + 1 < S) ? a.z[row + s + 1] : 0.f;
        pr = hn_part_row(a, ray, row, s, perm_lds);
        raw = (pr.second ? a.raw1 : a.raw)[pr.row];
        if (a.noise != nullptr) raw = __fadd_rn(raw, __fmul_rn(a.noise[row + s], a.noise_scale));
        if (!BACKWARD) {      // the sample's colour with the same round of loads (behind the weight store further down the
          const float* c = (pr.second ? a.rgb1 : a.rgb) + pr.row * 3;      // compiler cannot hoist it: one more latency)
          cr = c[0]; cg = c[1]; cb = c[2];
        }
      }
      const float dz = (s + 1 < S) ? (zn - zz) : last;
      const float dd = dz * dnorm;
      float sigma = a.variant == 0 ? hn_softplus(raw) : (a.variant == 1 ? fmaxf(raw, 0.0f) : raw);
      // filter_sigma (reference models.py:35-63): densities below the dust threshold and outside the bounding box
      // (a 0/1 mask the host builds from the sample points) are dropped; the factor is constant for the gradient
      float kfac = 1.0f;
      if (a.has_dust != 0 && !(sigma >= a.dust_threshold)) kfac = 0.0f;
      if (a.keep != nullptr && in) kfac *= a.keep[row + s];
      sigma *= kfac;
      float al = in ? (1.0f - expf(-sigma * dd)) : 0.0f;
      const float o = in ? (1.0f - al + eps) : 1.0f;
      // exclusive product scan: T_s = prod_{j<s} om_j
      const float inc = hn_wave_incl_scan_mul(o, lane);
      const float ex = hn_wave_shr1(inc, 1.0f);
      const float T = carry * ex;
      carry *= hn_wave_bcast(inc, 63);
      const float w = al * T;
      alpha[k] = al; om[k] = o; trans[k] = T; wgt[k] = w; pre[k] = raw; dist[k] = dd * kfac;   // d sigma' = kfac d sigma
      if (!BACKWARD) {
        if (in) {
          a.out_weights[row + s] = w;
          if (w_lds != nullptr) w_lds[s] = w;
          s_r += w * cr;
          s_g += w * cg;
          s_b += w * cb;
          s_d += w * zz;
          s_w += w;
          if (s < S - 1) s_wl += w;
        }
        if (a.out_med_depth != nullptr) {
          // first sample whose inclusive weight sum reaches 0.5 (model_utils.py:319-345)
          const float cs = csum + hn_wave_incl_scan_add(in ? w : 0.0f, lane);
          const unsigned long long m = __ballot(in && cs >= 0.5f);
          if (!med_found && m != 0ull) {
            const int first = __ffsll((long long)m) - 1;
            med_z = hn_wave_bcast(zz, first);
            const int sidx = k * 64 + first;
            if (a.warped != nullptr) {
              const HnPartRow mp = hn_part_row(a, ray, row, sidx, perm_lds);
              med_pt = (mp.second ? a.warped1 : a.warped)[mp.row * a.warped_ld];
            }
            med_found = true;
          }
          csum = hn_wave_bcast(cs, 63);
        }
      }
    }
  }
  if (!BACKWARD && a.out_warped != nullptr) {
    // the level's `warped_points` in sorted order, gathered from the parts: the ray's S x ld floats as ONE contiguous
    // run — lane-consecutive stores (a row per lane would scatter 28-byte pieces over the wave's store).  The merge
    // permutation is staged in LDS first (round 6): an element's source address then depends on no global load, and the
    // wave keeps eight gathers in flight — the loop used to pay two dependent memory latencies (perm, then the row) for
    // each of its S ld / 64 rounds: 10 of the launch's 15 us at config 2.
    const int ld = a.warped_ld, n = S * ld;
    float* __restrict__ wdst = a.out_warped + row * ld;
    const float* __restrict__ w0 = a.warped + (size_t)ray * a.split * ld;
    const float* __restrict__ w1 = a.warped1 + (size_t)ray * (S - a.split) * ld;
    if (perm_lds != nullptr) {
      for (int e0 = lane; e0 < n; e0 += 64 * 8) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int e = e0 + 64 * j;
          if (e < n) {
            const int s = e / ld, cc = e - s * ld, k = perm_lds[s];
            v[j] = k < a.split ? w0[k * ld + cc] : w1[(k - a.split) * ld + cc];
          }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (e0 + 64 * j < n) wdst[e0 + 64 * j] = v[j];
      }
    } else {
      for (int e = lane; e < n; e += 64) {
        const int s = e / ld, cc = e - s * ld;
        const HnPartRow pr = hn_part_row(a, ray, row, s);
        wdst[e] = (pr.second ? a.warped1 : a.warped)[pr.row * ld + cc];
      }
    }
  }
  if (!BACKWARD) {
    s_r = hn_wave_sum(s_r); s_g = hn_wave_sum(s_g); s_b = hn_wave_sum(s_b);
    s_d = hn_wave_sum(s_d); s_w = hn_wave_sum(s_w); s_wl = hn_wave_sum(s_wl);
    if (lane == 0) {
      float bg = a.white_bg ? (1.0f - s_w) : 0.0f;
      a.out_rgb[ray * 3 + 0] = s_r + bg;
      a.out_rgb[ray * 3 + 1] = s_g + bg;
      a.out_rgb[ray * 3 + 2] = s_b + bg;
      a.out_depth[ray] = s_d;
      a.out_acc[ray] = (a.variant != 1 && a.sample_at_infinity) ? s_wl : s_w;
      if (a.out_med_depth != nullptr) {
        a.out_med_depth[ray] = med_z;
        if (a.out_med_points != nullptr) {
          float first_pt = 0.0f;
          if (a.warped != nullptr) {
            const HnPartRow mp = hn_part_row(a, ray, row, 0, perm_lds);
            first_pt = (mp.second ? a.warped1 : a.warped)[mp.row * a.warped_ld];
          }
          a.out_med_points[ray] = med_found ? med_pt : first_pt;
        }
      }
    }
    return;
  }
  // ---- backward -------------------------------------------------------------------------------
  // L depends on w_s through rgb (sum w c), depth (sum w z), acc and explicit weight gradients;
  // white background adds -(g_r+g_g+g_b) to every dL/dw.
  float gr = 0.f, gg = 0.f, gb = 0.f, gd = 0.f, ga = 0.f;
  if (a.g_rgb != nullptr) { gr = a.g_rgb[ray * 3]; gg = a.g_rgb[ray * 3 + 1]; gb = a.g_rgb[ray * 3 + 2]; }
  if (a.g_depth != nullptr) gd = a.g_depth[ray];
  if (a.g_acc != nullptr) ga = a.g_acc[ray];
  const float gbg = a.white_bg ? -(gr + gg + gb) : 0.0f;
  const bool acc_drops_last = (a.variant != 1 && a.sample_at_infinity);
  float suffix = 0.0f;  // sum_{k > s} dL/dw_k * w_k  over later segments
#pragma unroll
  for (int k = HN_MAX_SEG - 1; k >= 0; --k) {
    if (k < nseg) {
      const int s = k * 64 + lane;
      const bool in = s < S;
      float dw = 0.0f, c0 = 0.f, c1 = 0.f, c2 = 0.f;
      HnPartRow pr{row, false};
      if (in) {
        pr = hn_part_row(a, ray, row, s, perm_lds);
        const float* c = (pr.second ? a.rgb1 : a.rgb) + pr.row * 3;
        c0 = c[0]; c1 = c[1]; c2 = c[2];
        dw = gr * c0 + gg * c1 + gb * c2 + gd * a.z[row + s] + gbg;
        if (!acc_drops_last || s < S - 1) dw += ga;
        if (a.g_weights != nullptr) dw += a.g_weights[row + s];
      }
      const float dww = in ? dw * wgt[k] : 0.0f;
      // exclusive suffix sum inside the segment: sum_{j > lane} dww_j
      float v = dww;
#pragma unroll
      for (int dlt = 1; dlt < 64; dlt <<= 1) {
        const float o = __shfl_down(v, dlt, 64);
        if (lane + dlt < 64) v += o;
      }
      const float seg_total = __shfl(v, 0, 64);
      const float after = (v - dww) + suffix;
      suffix += seg_total;
      if (in) {
        const float dalpha = dw * trans[k] - after / om[k];
        // alpha = 1 - exp(-sigma*dist)  ->  dalpha/dsigma = dist * exp(-sigma*dist) = dist * (1 - alpha)
        const float dsigma = dalpha * dist[k] * (1.0f - alpha[k]);
        float draw;
        if (a.variant == 0) draw = dsigma * (pre[k] > 20.0f ? 1.0f : hn_sigmoid(pre[k]));
        else if (a.variant == 1) draw = pre[k] > 0.0f ? dsigma : 0.0f;
        else draw = dsigma;
        (pr.second ? a.d_raw1 : a.d_raw)[pr.row] = draw;
        const float ws = wgt[k];
        float* dc = (pr.second ? a.d_rgb1 : a.d_rgb) + pr.row * 3;
        dc[0] = gr * ws;
        dc[1] = gg * ws;
        dc[2] = gb * ws;
      }
    }
  }
}
template <bool BACKWARD>
__global__ __launch_bounds__(256) void hn_composite_kernel(const HnCompositeArgs a) {
  __shared__ int s_perm[4][64 * HN_MAX_SEG];
  const int lane = threadIdx.x & 63;
  const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= a.n_rays) return;
  hn_composite_ray<BACKWARD>(a, ray, lane, nullptr, s_perm[threadIdx.x >> 6]);
}

static int hn_check_comp(const HnCompositeArgs* a, bool bwd) {
  if (a == nullptr) return -1;
  if (a->n_rays <= 0 || a->n_samples <= 0 || a->n_samples > 64 * HN_MAX_SEG) return -2;
  if (a->rgb == nullptr || a->raw == nullptr || a->z == nullptr || a->dirs == nullptr) return -3;
  if (!bwd && (a->out_rgb == nullptr || a->out_depth == nullptr || a->out_acc == nullptr || a->out_weights == nullptr))
    return -3;
  if (bwd && (a->d_rgb == nullptr || a->d_raw == nullptr)) return -3;
  if (a->perm != nullptr) {     // a level in two parts
    if (a->split < 0 || a->split > a->n_samples) return -2;
    if (a->split < a->n_samples && (a->rgb1 == nullptr || a->raw1 == nullptr)) return -3;
    if (bwd && a->split < a->n_samples && (a->d_rgb1 == nullptr || a->d_raw1 == nullptr)) return -3;
    if (a->warped != nullptr && a->split < a->n_samples && a->warped1 == nullptr) return -3;
  }
  if (!bwd && a->out_warped != nullptr && a->warped == nullptr) return -3;
  return 0;
}

extern "C" int hn_composite_forward(const HnCompositeArgs* a, hnStream_t stream) {
  int rc = hn_check_comp(a, false);
  if (rc) return rc;
  hipLaunchKernelGGL(hn_composite_kernel<false>, dim3((a->n_rays + 3) / 4), dim3(256), 0, (hipStream_t)stream, *a);
  HN_CHECK_LAUNCH();
  return 0;
}
extern "C" int hn_composite_backward(const HnCompositeArgs* a, hnStream_t stream) {
  int rc = hn_check_comp(a, true);
  if (rc) return rc;
  hipLaunchKernelGGL(hn_composite_kernel<true>, dim3((a->n_rays + 3) / 4), dim3(256), 0, (hipStream_t)stream, *a);
  HN_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// median depth from given weights: model_utils.compute_opaqueness_mask / compute_depth_index / compute_depth_map
// (hypernerf/model_utils.py:319-362).  One wave per ray; the same inclusive scan + ballot the compositing kernel
// uses for its own `med_depth`.  mask[b, s] = 1 at the FIRST sample whose inclusive weight sum reaches the
// threshold (0 everywhere if none does), index = argmax(mask) (0 if none), depth = sum(mask * z).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void hn_depth_index_kernel(const float* __restrict__ w, const float* __restrict__ z,
                                                             int n_rays, int S, float thr, int64_t* out_idx,
                                                             float* out_depth, float* out_mask) {
  const int lane = threadIdx.x & 63;
  const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= n_rays) return;
  const size_t row = (size_t)ray * S;
  float csum = 0.0f;
  int found = -1;
  for (int k = 0; k * 64 < S; ++k) {
    const int s = k * 64 + lane;
    const bool in = s < S;
    const float cs = csum + hn_wave_incl_scan_add(in ? w[row + s] : 0.0f, lane);
    const unsigned long long m = __ballot(in && cs >= thr);
    if (found < 0 && m != 0ull) found = k * 64 + __ffsll((long long)m) - 1;
    csum = hn_wave_bcast(cs, 63);
  }
  if (out_mask != nullptr)
    for (int s = lane; s < S; s += 64) out_mask[row + s] = s == found ? 1.0f : 0.0f;
  if (lane == 0) {
    if (out_idx != nullptr) out_idx[ray] = found < 0 ? 0 : found;
    if (out_depth != nullptr) out_depth[ray] = (found >= 0 && z != nullptr) ? z[row + found] : 0.0f;
  }
}

extern "C" int hn_depth_index(const float* weights, const float* z, int n_rays, int n_samples, float threshold,
                              int64_t* out_index, float* out_depth, float* out_mask, hnStream_t stream) {
  if (n_rays < 0 || n_samples <= 0) return -2;
  if (n_rays == 0) return 0;
  if (weights == nullptr || (out_depth != nullptr && z == nullptr)) return -3;
  hipLaunchKernelGGL(hn_depth_index_kernel, dim3((n_rays + 3) / 4), dim3(256), 0, (hipStream_t)stream, weights, z,
                     n_rays, n_samples, threshold, out_index, out_depth, out_mask);
  HN_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// inverse-CDF sampling + merge sort (one wave per ray; LDS: cdf/bins + 512-entry sort buffer)
// ------------------------------------------------------------------------------------------------
constexpr int HN_PDF_MAXC = 256;   // max coarse samples
constexpr int HN_PDF_MAXT = 512;   // max coarse + fine

// One ray's inverse-CDF sampling + merge on one wave.  `wr`: the ray's bin weights (global memory, or LDS when the same
// wave has just composited them); cdf / bins / srt / sidx: this wave's LDS scratch.
struct HnPdfArgs {
  const float* weights; int w_ld;
  const float* bins_in; int nb;
  const float* z; int nc;
  const float* u; const float* origins; const float* dirs; int ray_ld, n_rays, nf;
  float* z_all; float* pts; int64_t* inds; float* z_samples; int32_t* perm; float* pts_new;
};
template <bool PERM>
HN_DEV void hn_pdf_ray(const HnPdfArgs& q, int ray, int lane, const float* wr, float* cdf, float* bins, float* srt, int* sidx) {
  const float* __restrict__ bins_in = q.bins_in;
  const float* __restrict__ z = q.z;
  const float* __restrict__ u = q.u;
  const float* __restrict__ origins = q.origins;
  const float* __restrict__ dirs = q.dirs;
  float* __restrict__ z_all = q.z_all;
  float* __restrict__ pts = q.pts;
  int64_t* __restrict__ inds = q.inds;
  float* __restrict__ z_samples = q.z_samples;
  int32_t* __restrict__ perm = q.perm;
  float* __restrict__ pts_new = q.pts_new;
  const int nb = q.nb, nc = q.nc, nf = q.nf, ray_ld = q.ray_ld;
  const int ncdf = nb + 1;     // cdf entries = bin edges
  const float* zr = z != nullptr ? z + (size_t)ray * nc : nullptr;
  if (bins_in != nullptr) {
    for (int i = lane; i < ncdf; i += 64) bins[i] = bins_in[(size_t)ray * ncdf + i];
  } else {
    for (int i = lane; i < ncdf; i += 64) bins[i] = 0.5f * (zr[i + 1] + zr[i]);   // z_vals_mid
  }
  for (int i = lane; i < nb; i += 64) srt[i] = wr[i] + 1e-5f;
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_wave_barrier();
  {
    // Normaliser = fp64 sum rounded once, cdf = fp64 prefix sums rounded per entry (what the oracle defines; ATen's
    // fp32 cascade is ISA dependent).  Both run over the whole wave: every term is an fp32 value, and fp64 sums of
    // <= 255 fp32 values whose magnitudes lie within a factor 2^21 of each other are EXACT (24 + 8 + 21 = 53 bits),
    // i.e. independent of the summation order — weights + 1e-5 span [1e-5, ~1], a factor 2^17.  (Beyond that range
    // the result can differ from the sequential sum by one fp64 ulp before the rounding to fp32.)
    const int per = (nb + 63) >> 6;                 // consecutive entries per lane (<= 4)
    const int i0 = lane * per;
    double loc = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (k < per && i0 + k < nb) loc += (double)srt[i0 + k];
    double tot = loc;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) tot += __shfl_xor(tot, m, 64);
    const float norm = (float)tot;
    float pdf[4];
    double run = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      pdf[k] = (k < per && i0 + k < nb) ? __fdiv_rn(srt[i0 + k], norm) : 0.0f;
      run += (double)pdf[k];
    }
    double incl = run;                               // inclusive scan of the lanes' sums
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const double o = __shfl_up(incl, d, 64);
      if (lane >= d) incl += o;
    }
    double base = incl - run;                        // sum of all earlier lanes' entries (exact, see above)
    if (lane == 0) cdf[0] = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (k < per && i0 + k < nb) {
        base += (double)pdf[k];
        cdf[i0 + k + 1] = (float)base;
      }
    }
  }
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_wave_barrier();
  const int nmerge = zr != nullptr ? nc : 0;
  const int total = nmerge + nf;
  for (int i = lane; i < nf; i += 64) {
    const float uu = u[(size_t)ray * nf + i];
    // searchsorted(cdf, u, right=True): number of entries <= u
    int lo = 0, hi = ncdf;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (cdf[mid] <= uu) lo = mid + 1; else hi = mid;
    }
    const int ind = lo;
    const int below = ind - 1 < 0 ? 0 : ind - 1;
    const int above = ind > nb ? nb : ind;
    const float c0 = cdf[below], c1 = cdf[above];
    const float b0 = bins[below], b1 = bins[above];
    float den = __fsub_rn(c1, c0);
    if (den < 1e-5f) den = 1.0f;
    const float t = __fdiv_rn(__fsub_rn(uu, c0), den);
    const float smp = __fadd_rn(b0, __fmul_rn(t, __fsub_rn(b1, b0)));
    if (inds != nullptr) inds[(size_t)ray * nf + i] = ind;
    if (z_samples != nullptr) z_samples[(size_t)ray * nf + i] = smp;
    if (pts_new != nullptr) {
      const float* o = origins + (size_t)ray * ray_ld;
      const float* d = dirs + (size_t)ray * ray_ld;
#pragma unroll
      for (int c = 0; c < 3; ++c) pts_new[((size_t)ray * nf + i) * 3 + c] = __fadd_rn(o[c], __fmul_rn(smp, d[c]));
    }
    srt[nmerge + i] = smp;
  }
  if (z_all == nullptr) return;
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_wave_barrier();
  for (int i = lane; i < nmerge; i += 64) srt[i] = zr[i];
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_wave_barrier();
  // Merge by RANK (round 6) when the level's own depths arrive sorted and every value is a number — what the render path
  // always hands over (stratified / previous-level depths): the sorted position of an entry of cat(z, z_samples) is the
  // number of entries before it in (depth, position) order, which each lane COUNTS for its own entries — a coarse depth k
  // stands behind k coarse depths and behind the new samples below it; new sample i behind the coarse depths <= it
  // (binary search) and the new samples below it or equal with a smaller index — then scatters them to that position in
  // LDS.  The same sequence, bit for bit, as the (depth, position) bitonic sort below — which stays for any other input —
  // in 2 LDS passes instead of log2(n)(log2(n)+1)/2 = 28 rounds of compare-exchange at 128 entries (16 -> 10 us a launch).
  bool fast = true;
  for (int i = lane; i < total; i += 64) {
    const float x = srt[i];
    fast = fast && (x == x) && !(i + 1 < nmerge && !(x <= srt[i + 1]));
  }
  if (__ballot(!fast) == 0ull) {
    float xs[12];
    int rk[12];
#pragma unroll
    for (int k = 0; k < 4; ++k) {          // this lane's coarse depths: entries lane + 64 k
      const int i = lane + 64 * k;
      if (i < nmerge) {
        const float x = srt[i];
        int c = i;
        for (int j = 0; j < nf; ++j) c += srt[nmerge + j] < x ? 1 : 0;
        xs[k] = x; rk[k] = c;
      }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {          // this lane's new samples
      const int i = lane + 64 * k;
      if (i < nf) {
        const float x = srt[nmerge + i];
        int lo = 0, hi = nmerge;           // coarse depths <= x (they all stand at smaller positions: ties count)
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (srt[mid] <= x) lo = mid + 1; else hi = mid;
        }
        int c = lo;
        for (int j = 0; j < nf; ++j) {
          const float y = srt[nmerge + j];
          c += (y < x || (y == x && j < i)) ? 1 : 0;
        }
        xs[4 + k] = x; rk[4 + k] = c;
      }
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();       // every lane has read what it needs: the buffer becomes the output
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (lane + 64 * k < nmerge) { srt[rk[k]] = xs[k]; if (PERM) sidx[rk[k]] = lane + 64 * k; }
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (lane + 64 * k < nf) { srt[rk[4 + k]] = xs[4 + k]; if (PERM) sidx[rk[4 + k]] = nmerge + lane + 64 * k; }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
  } else {
  // pad to a power of two with +inf and bitonic-sort ascending
  int n2 = 1;
  while (n2 < total) n2 <<= 1;
  for (int i = total + lane; i < n2; i += 64) srt[i] = __builtin_inff();
  if (PERM)
    for (int i = lane; i < n2; i += 64) sidx[i] = i;
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_wave_barrier();
  for (int k = 2; k <= n2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = lane; i < n2; i += 64) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const float x = srt[i], y = srt[ixj];
          const bool up = (i & k) == 0;
          if (PERM) {
            // (depth, position) pairs: equal depths keep the order of cat(z, z_samples) — the key sequence that
            // comes out is the same as without the payload
            const int px = sidx[i], py = sidx[ixj];
            const bool gt = x > y || (x == y && px > py);
            if (gt == up) { srt[i] = y; srt[ixj] = x; sidx[i] = py; sidx[ixj] = px; }
          } else if ((x > y) == up) { srt[i] = y; srt[ixj] = x; }
        }
      }
      __builtin_amdgcn_s_waitcnt(0);
      __builtin_amdgcn_wave_barrier();
    }
  }
  }
  for (int i = lane; i < total; i += 64) {
    const float zz = srt[i];
    z_all[(size_t)ray * total + i] = zz;
    if (PERM) perm[(size_t)ray * total + i] = sidx[i];
    if (pts != nullptr) {
      const float* o = origins + (size_t)ray * ray_ld;
      const float* d = dirs + (size_t)ray * ray_ld;
#pragma unroll
      for (int c = 0; c < 3; ++c) pts[((size_t)ray * total + i) * 3 + c] = __fadd_rn(o[c], __fmul_rn(zz, d[c]));
    }
  }
}
template <bool PERM>
__global__ __launch_bounds__(256) void hn_sample_pdf_kernel(const HnPdfArgs q) {
  __shared__ float s_cdf[4][HN_PDF_MAXC];
  __shared__ float s_bins[4][HN_PDF_MAXC];
  __shared__ float s_sort[4][HN_PDF_MAXT];
  __shared__ int s_idx[PERM ? 4 : 1][PERM ? HN_PDF_MAXT : 1];   // the sort's payload: position in cat(z, z_samples)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int ray = blockIdx.x * 4 + wv;
  if (ray >= q.n_rays) return;
  hn_pdf_ray<PERM>(q, ray, lane, q.weights + (size_t)ray * q.w_ld, s_cdf[wv], s_bins[wv], s_sort[wv], s_idx[PERM ? wv : 0]);
}
// A level's compositing and the inverse-CDF sampling of the NEXT level from its weights as one launch (round 6; reference
// call order models.py:744-768: render_samples(coarse) -> sample_pdf(coarse weights)): the same wave composites a ray and
// then draws its fine samples — the weights reach the sampler through LDS (they are also written out: the level's
// `weights` output), one dispatch and one HBM round trip of (B, S) floats fewer.  Same arithmetic as the two kernels.
template <bool PERM>
__global__ __launch_bounds__(256) void hn_composite_pdf_kernel(const HnCompositeArgs a, const HnPdfArgs q) {
  __shared__ float s_cdf[4][HN_PDF_MAXC];
  __shared__ float s_bins[4][HN_PDF_MAXC];
  __shared__ float s_sort[4][HN_PDF_MAXT];
  __shared__ float s_w[4][HN_PDF_MAXC + 1];
  __shared__ int s_idx[PERM ? 4 : 1][PERM ? HN_PDF_MAXT : 1];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int ray = blockIdx.x * 4 + wv;
  if (ray >= a.n_rays) return;
  hn_composite_ray<false>(a, ray, lane, s_w[wv]);
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_wave_barrier();
  hn_pdf_ray<PERM>(q, ray, lane, s_w[wv] + 1, s_cdf[wv], s_bins[wv], s_sort[wv], s_idx[PERM ? wv : 0]);
}

static int hn_pdf_check(const HnPdfArgs& q, bool own_weights) {
  if (q.n_rays <= 0 || q.nb < 1 || q.nf <= 0) return -2;
  if (q.z != nullptr && q.nc < 2) return -2;
  if (q.bins_in == nullptr && (q.z == nullptr || q.nc - 2 != q.nb)) return -2;
  if (q.nb + 1 > HN_PDF_MAXC || (q.z != nullptr ? q.nc : 0) + q.nf > HN_PDF_MAXT) return -2;
  if ((own_weights && q.weights == nullptr) || q.u == nullptr) return -3;
  if (q.z_all == nullptr && q.z_samples == nullptr && q.inds == nullptr && q.pts_new == nullptr) return -3;
  if (q.pts != nullptr && (q.origins == nullptr || q.dirs == nullptr || q.z_all == nullptr)) return -3;
  if (q.pts_new != nullptr && (q.origins == nullptr || q.dirs == nullptr)) return -3;
  if (q.perm != nullptr && (q.z == nullptr || q.z_all == nullptr)) return -3;
  return 0;
}

extern "C" int hn_sample_pdf_split(const float* weights, int w_ld, const float* bins, int n_bins, const float* z,
                                   int n_coarse, const float* u, const float* origins, const float* dirs, int ray_ld,
                                   int n_rays, int n_fine, float* z_all, float* pts, int64_t* inds, float* z_samples,
                                   int32_t* perm, float* pts_new, hnStream_t stream) {
  const HnPdfArgs q = {weights, w_ld, bins, n_bins, z, n_coarse, u, origins, dirs, ray_ld, n_rays, n_fine,
                       z_all, pts, inds, z_samples, perm, pts_new};
  const int rc = hn_pdf_check(q, true);
  if (rc) return rc;
  if (perm != nullptr)
    hipLaunchKernelGGL(hn_sample_pdf_kernel<true>, dim3((n_rays + 3) / 4), dim3(256), 0, (hipStream_t)stream, q);
  else
    hipLaunchKernelGGL(hn_sample_pdf_kernel<false>, dim3((n_rays + 3) / 4), dim3(256), 0, (hipStream_t)stream, q);
  HN_CHECK_LAUNCH();
  return 0;
}

extern "C" int hn_composite_sample_pdf(const HnCompositeArgs* a, const float* u, const float* origins, const float* dirs,
                                       int ray_ld, int n_fine, float* z_all, float* pts, int64_t* inds, float* z_samples,
                                       int32_t* perm, float* pts_new, hnStream_t stream) {
  int rc = hn_check_comp(a, false);
  if (rc) return rc;
  if (a->perm != nullptr) return -2;           // the level that feeds the sampler is composited in one part
  // hn_sample_pdf's fused form: bins = midpoints of the level's z, weights = columns 1 .. S-2 of the level's weights
  const HnPdfArgs q = {nullptr, 0, nullptr, a->n_samples - 2, a->z, a->n_samples, u, origins, dirs, ray_ld, a->n_rays,
                       n_fine, z_all, pts, inds, z_samples, perm, pts_new};
  rc = hn_pdf_check(q, false);
  if (rc) return rc;
  if (perm != nullptr)
    hipLaunchKernelGGL(hn_composite_pdf_kernel<true>, dim3((a->n_rays + 3) / 4), dim3(256), 0, (hipStream_t)stream, *a, q);
  else
    hipLaunchKernelGGL(hn_composite_pdf_kernel<false>, dim3((a->n_rays + 3) / 4), dim3(256), 0, (hipStream_t)stream, *a, q);
  HN_CHECK_LAUNCH();
  return 0;
}

extern "C" int hn_sample_pdf(const float* weights, int w_ld, const float* bins, int n_bins, const float* z,
                             int n_coarse, const float* u, const float* origins, const float* dirs, int ray_ld,
                             int n_rays, int n_fine, float* z_all, float* pts, int64_t* inds, float* z_samples,
                             hnStream_t stream) {
  if (z_all == nullptr && z_samples == nullptr && inds == nullptr) return -3;
  return hn_sample_pdf_split(weights, w_ld, bins, n_bins, z, n_coarse, u, origins, dirs, ray_ld, n_rays, n_fine, z_all,
                             pts, inds, z_samples, nullptr, nullptr, stream);
}

// ------------------------------------------------------------------------------------------------
// GLO embedding
// ------------------------------------------------------------------------------------------------
__global__ void hn_embed_gather_kernel(const float* __restrict__ table, const int64_t* __restrict__ idx, int n_rays,
                                       int dim, int n_rows, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rays * dim) return;
  const int b = i / dim, c = i % dim;
  // an index outside the table poisons its row with NaN (nn.Embedding raises a device assert there; a silent clamp
  // would train the wrong row): the loss goes NaN at once instead of rendering a wrong image
  const int64_t row = idx[b];
  out[i] = (row < 0 || row >= n_rows) ? __builtin_nanf("") : table[row * dim + c];
}

__global__ __launch_bounds__(256) void hn_embed_bwd_kernel(const float* __restrict__ d_points, int ld, int col0,
                                                           const int64_t* __restrict__ idx, int n_rays, int S,
                                                           int dim, int n_rows, float* __restrict__ d_table) {
  const int lane = threadIdx.x & 63;
  const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= n_rays) return;
  const int64_t row = idx[ray];
  if (row < 0 || row >= n_rows) return;       // no row to receive it (the forward already produced NaN)
  for (int c = 0; c < dim; ++c) {
    float v = 0.0f;
    for (int s = lane; s < S; s += 64) v += d_points[((size_t)ray * S + s) * ld + col0 + c];
    v = hn_wave_sum(v);
    if (lane == 0) atomicAdd(d_table + row * dim + c, v);
  }
}

extern "C" int hn_embed_gather(const float* table, const int64_t* idx, int n_rays, int dim, int n_rows, float* out,
                               hnStream_t stream) {
  if (n_rays <= 0 || dim <= 0 || n_rows <= 0) return -2;
  if (table == nullptr || idx == nullptr || out == nullptr) return -3;
  hipLaunchKernelGGL(hn_embed_gather_kernel, dim3((n_rays * dim + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     table, idx, n_rays, dim, n_rows, out);
  HN_CHECK_LAUNCH();
  return 0;
}

extern "C" int hn_embed_backward(const float* d_points, int ld, int col0, const int64_t* idx, int n_rays,
                                 int n_samples, int dim, int n_rows, float* d_table, hnStream_t stream) {
  if (n_rays <= 0 || dim <= 0 || n_rows <= 0 || n_samples <= 0) return -2;
  if (d_points == nullptr || idx == nullptr || d_table == nullptr) return -3;
  hipLaunchKernelGGL(hn_embed_bwd_kernel, dim3((n_rays + 3) / 4), dim3(256), 0, (hipStream_t)stream, d_points, ld,
                     col0, idx, n_rays, n_samples, dim, n_rows, d_table);
  HN_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// On-device ray generation (datasets/ray_utils.py:5-93 + the row layout of datasets/llff.py:244-264):
// pixel (col i, row j) -> camera direction ((i - W/2)/f, -(j - H/2)/f, -1) -> world (c2w 3x4) -> normalised;
// origin = c2w[:, 3]; optional NDC transform (near plane 1.0 in the reference's call); one thread per pixel writes
// the whole (8|9)-float ray row [o, d, near, far(, image id)] — 36 B/pixel, HBM bound.
// ------------------------------------------------------------------------------------------------
__global__ void hn_generate_rays_kernel(int H, int W, float focal, const float* c2w, int ndc, float ndc_near,
                                        float near, float far, float image_id, int row_floats, float* rays) {
  const int pix = blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= H * W) return;
  const int j = pix / W, i = pix - j * W;
  const float dx = __fdiv_rn(__fsub_rn((float)i, __fdiv_rn((float)W, 2.0f)), focal);
  const float dy = -__fdiv_rn(__fsub_rn((float)j, __fdiv_rn((float)H, 2.0f)), focal);
  const float dz = -1.0f;
  float d[3], o[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    d[k] = __fadd_rn(__fadd_rn(__fmul_rn(dx, c2w[4 * k]), __fmul_rn(dy, c2w[4 * k + 1])), __fmul_rn(dz, c2w[4 * k + 2]));
    o[k] = c2w[4 * k + 3];
  }
  const float nrm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(d[0], d[0]), __fmul_rn(d[1], d[1])), __fmul_rn(d[2], d[2])));
#pragma unroll
  for (int k = 0; k < 3; ++k) d[k] = __fdiv_rn(d[k], nrm);
  if (ndc) {   // get_ndc_rays, ray_utils.py:52-93
    const float t = __fdiv_rn(-__fadd_rn(ndc_near, o[2]), d[2]);
#pragma unroll
    for (int k = 0; k < 3; ++k) o[k] = __fadd_rn(o[k], __fmul_rn(t, d[k]));
    const float ox_oz = __fdiv_rn(o[0], o[2]), oy_oz = __fdiv_rn(o[1], o[2]);
    const float sx = __fdiv_rn(-1.0f, __fdiv_rn((float)W, __fmul_rn(2.0f, focal)));
    const float sy = __fdiv_rn(-1.0f, __fdiv_rn((float)H, __fmul_rn(2.0f, focal)));
    const float o0 = __fmul_rn(sx, ox_oz), o1 = __fmul_rn(sy, oy_oz);
    const float o2 = __fadd_rn(1.0f, __fdiv_rn(__fmul_rn(2.0f, ndc_near), o[2]));
    const float d0 = __fmul_rn(sx, __fsub_rn(__fdiv_rn(d[0], d[2]), ox_oz));
    const float d1 = __fmul_rn(sy, __fsub_rn(__fdiv_rn(d[1], d[2]), oy_oz));
    const float d2 = __fsub_rn(1.0f, o2);
    o[0] = o0; o[1] = o1; o[2] = o2; d[0] = d0; d[1] = d1; d[2] = d2;
  }
  float* r = rays + (size_t)pix * row_floats;
  r[0] = o[0]; r[1] = o[1]; r[2] = o[2]; r[3] = d[0]; r[4] = d[1]; r[5] = d[2]; r[6] = near; r[7] = far;
  if (row_floats > 8) r[8] = image_id;
}

extern "C" int hn_generate_rays(int H, int W, float focal, const float* c2w, int ndc, float ndc_near, float near,
                                float far, float image_id, int row_floats, float* rays, hnStream_t stream) {
  if (H <= 0 || W <= 0 || !(focal > 0.0f) || (row_floats != 8 && row_floats != 9)) return -2;
  if (c2w == nullptr || rays == nullptr) return -3;
  const int n = H * W;
  hipLaunchKernelGGL(hn_generate_rays_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, H, W, focal,
                     c2w, ndc, ndc_near, near, far, image_id, row_floats, rays);
  HN_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Fused Adam over a flat parameter arena (torch.optim.Adam semantics, utils.get_optimizer's default):
// one pass over p, g, m, v (28 B/parameter, HBM bound), the step counter lives on the device so the launch can be
// captured in a HIP graph, and the gradient is zeroed on the way out (saves the separate fill of the next step).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void hn_adam_kernel(float* p, float* g, float* m, float* v, long long n,
                                                       const float* __restrict__ hyper, float* step,
                                                       int zero_grad) {
  // every block reads step[0] (hn_adam_consts) before it does anything else; the block that finishes LAST advances it
  const HnAdamConsts k = hn_adam_consts(hyper, step);
  const long long stride = (long long)gridDim.x * blockDim.x * 4;
  for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 4 <= n) {
      f32x4 pp = *reinterpret_cast<f32x4*>(p + i), gg = *reinterpret_cast<f32x4*>(g + i);
      f32x4 mm = *reinterpret_cast<f32x4*>(m + i), vv = *reinterpret_cast<f32x4*>(v + i);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float pe = pp[e], me = mm[e], ve = vv[e];
        hn_adam_update(k, pe, gg[e], me, ve);
        pp[e] = pe; mm[e] = me; vv[e] = ve;
      }
      *reinterpret_cast<f32x4*>(p + i) = pp;
      *reinterpret_cast<f32x4*>(m + i) = mm;
      *reinterpret_cast<f32x4*>(v + i) = vv;
      if (zero_grad) *reinterpret_cast<f32x4*>(g + i) = f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
      for (long long j = i; j < n; ++j) {
        hn_adam_update(k, p[j], g[j], m[j], v[j]);
        if (zero_grad) g[j] = 0.f;
      }
    }
  }
  hn_adam_ticket(step, k.t);
}

extern "C" int hn_adam_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, long long n,
                            const float* hyper_dev, float* step_dev, int zero_grad, hnStream_t stream) {
  if (n <= 0) return -2;
  if (params == nullptr || grads == nullptr || exp_avg == nullptr || exp_avg_sq == nullptr || step_dev == nullptr ||
      hyper_dev == nullptr)
    return -3;
  if ((((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) != 0) return -4;
  // one block per CU, grid-stride: every thread pays the bias-correction arithmetic (two powf, an rsqrtf) once for ~6
  // vectors instead of once per vector (2048 blocks: 26.3 us per launch at config 2, 512: 16.5, 256: 15.2)
  long long blocks = (n / 4 + 255) / 256;
  if (blocks > 256) blocks = 256;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(hn_adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg,
                     exp_avg_sq, n, hyper_dev, step_dev, zero_grad);
  HN_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// random draws of a render step in ONE launch: the reference draws t_rand ~ U[0,1) (model_utils.py:31), the density
// noise ~ N(0,1) per level (model_utils.py:300-317) and u ~ U[0,1) (model_utils.py:226) with four ATen launches; here
// up to HN_MAX_DRAWS buffers are filled by one counter-based generator (Philox4x32-10, the generator torch itself
// uses on the GPU): thread i of the launch encrypts counter (offset + i, buffer id) under the key `seed` into four
// 32-bit words = four uniforms ((x >> 8) * 2^-24, as torch.rand) or two Box-Muller pairs.  seed / offset / ticket
// live in device memory (state[0..2]): every block reads the offset first, the block that finishes last advances it
// by the number of threads — so a captured launch draws fresh numbers on every HIP-graph replay.
// ------------------------------------------------------------------------------------------------
struct HnDrawTable {
  float* ptr[HN_MAX_DRAWS];
  long long n[HN_MAX_DRAWS];
  long long first_thread[HN_MAX_DRAWS + 1];    // prefix sums of ceil(n / 4)
  int kind[HN_MAX_DRAWS];                      // 0 = uniform [0,1), 1 = standard normal
  int count;
};

HN_DEV void hn_philox_round(uint32_t (&c)[4], const uint32_t (&k)[2]) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k[0], n1 = (uint32_t)p1;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k[1], n3 = (uint32_t)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
HN_DEV void hn_philox4x32_10(uint32_t (&c)[4], uint64_t seed) {
  uint32_t k[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    hn_philox_round(c, k);
    k[0] += 0x9E3779B9u;
    k[1] += 0xBB67AE85u;
  }
}

// Coarse sampling riding on the draw of t_rand (hn_render_prologue): the thread that has just drawn four values of the
// (B, n) uniform buffer places those four samples — hn_sample_kernel's arithmetic, operation for operation.
struct HnSampleRide {
  int draw;                 // index of the uniform draw that is t_rand, -1: none
  int n_rays, n, ray_ld, per_ray_bounds;
  float scale;
  const float* origins; const float* dirs; const float* lower; const float* upper;
  float* z_out; float* pts_out;
};
HN_DEV void hn_place_sample(const HnSampleRide& sp, long long i, float t) {
  const int b = (int)(i / sp.n), s = (int)(i % sp.n);
  const size_t bi = sp.per_ray_bounds ? (size_t)i : (size_t)s;
  float z = sp.lower[bi];
  // z = lower + (upper - lower) * (scale * t): three separately rounded fp32 ops, as ATen does
  if (sp.scale != 1.0f) t = __fmul_rn(sp.scale, t);
  z = __fadd_rn(z, __fmul_rn(__fsub_rn(sp.upper[bi], z), t));
  sp.z_out[i] = z;
  if (sp.pts_out != nullptr) {
#pragma unroll
    for (int c = 0; c < 3; ++c)
      sp.pts_out[i * 3 + c] = __fadd_rn(sp.origins[(size_t)b * sp.ray_ld + c], __fmul_rn(z, sp.dirs[(size_t)b * sp.ray_ld + c]));
  }
}

// the draws of blocks [blk, blk + nblk) of a launch section of `nblk` blocks (grid-stride inside the section); the block
// that finishes last advances the generator's offset
template <bool RIDE>
HN_DEV void hn_random_section(const HnDrawTable& t, unsigned long long* state, int blk, int nblk, const HnSampleRide& sp) {
  const unsigned long long seed = state[0], offset = state[1];
  const long long total = t.first_thread[t.count];
  for (long long i = (long long)blk * blockDim.x + threadIdx.x; i < total; i += (long long)nblk * blockDim.x) {
    int b = 0;
#pragma unroll
    for (int k = 1; k < HN_MAX_DRAWS; ++k)
      if (k < t.count && i >= t.first_thread[k]) b = k;
    const long long j = (i - t.first_thread[b]) * 4;
    const unsigned long long ctr = offset + (unsigned long long)i;
    uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), (uint32_t)b, 0x484e3033u};
    hn_philox4x32_10(c, seed);
    float v[4];
    if (t.kind[b] == 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (float)(c[e] >> 8) * 5.9604644775390625e-8f;           // [0, 1)
    } else {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const float u1 = ((float)(c[2 * e] >> 8) + 1.0f) * 5.9604644775390625e-8f;              // (0, 1]
        const float u2 = (float)(c[2 * e + 1] >> 8) * 5.9604644775390625e-8f;
        const float r = sqrtf(-2.0f * logf(u1));
        float sn, cs;
        sincosf(6.283185307179586f * u2, &sn, &cs);
        v[2 * e] = r * cs;
        v[2 * e + 1] = r * sn;
      }
    }
    float* dst = t.ptr[b] + j;
    if (j + 4 <= t.n[b] && (((uintptr_t)dst) & 15) == 0) {
      *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
    } else {
      for (int e = 0; e < 4 && j + e < t.n[b]; ++e) dst[e] = v[e];
    }
    if (RIDE && b == sp.draw)
      for (int e = 0; e < 4 && j + e < t.n[b]; ++e) hn_place_sample(sp, j + e, v[e]);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned* ticket = reinterpret_cast<unsigned*>(state + 2);
    if (atomicAdd(ticket, 1u) == (unsigned)nblk - 1) {      // every block of the section has read the old offset by now
      state[1] = offset + (unsigned long long)total;
      *ticket = 0u;
    }
  }
}

__global__ __launch_bounds__(256) void hn_random_kernel(const HnDrawTable t, unsigned long long* state) {
  hn_random_section<false>(t, state, (int)blockIdx.x, (int)gridDim.x, HnSampleRide{});
}

static int hn_draw_table_fill(const HnDraw* draws_host, int n_draws, HnDrawTable& t, long long& blocks) {
  blocks = 0;
  if (n_draws < 0 || n_draws > HN_MAX_DRAWS) return -1;
  if (n_draws == 0) return 0;
  if (draws_host == nullptr) return -3;
  long long threads = 0;
  for (int i = 0; i < n_draws; ++i) {
    if (draws_host[i].n < 0 || (draws_host[i].kind != 0 && draws_host[i].kind != 1)) return -2;
    if (draws_host[i].n > 0 && draws_host[i].ptr == nullptr) return -3;
    t.ptr[i] = draws_host[i].ptr; t.n[i] = draws_host[i].n; t.kind[i] = draws_host[i].kind;
    t.first_thread[i] = threads;
    threads += (draws_host[i].n + 3) / 4;
  }
  t.count = n_draws;
  t.first_thread[n_draws] = threads;
  blocks = (threads + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  return 0;
}

extern "C" int hn_random_fill(const HnDraw* draws_host, int n_draws, uint64_t* state_dev, hnStream_t stream) {
  HnDrawTable t = {};
  long long blocks = 0;
  const int rc = hn_draw_table_fill(draws_host, n_draws, t, blocks);
  if (rc != 0) return rc;
  if (n_draws == 0 || blocks == 0) return 0;
  if (state_dev == nullptr) return -3;
  hipLaunchKernelGGL(hn_random_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, t,
                     (unsigned long long*)state_dev);
  HN_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// The head of a render step as ONE launch (round 6): everything the step needs before its first machine launch and that
// depends on nothing but the step's inputs — the weight streams of its programs (hn_pack_units_multi), all random draws
// (hn_random_fill), the coarse samples placed from the t_rand draw (hn_sample_along_rays) and the int64 image ids of the
// ray rows (the reference's `.type(torch.long)`, model_utils.py:389-392).  Four launches of 5-13 us each, of which ~5 us
// apiece is the floor of a dispatch, become one: sections of the grid [pack | draws + samples | ids].
// ------------------------------------------------------------------------------------------------
struct HnPrologueDev {
  HnPackTable pack;
  HnDrawTable draws;
  HnSampleRide ride;
  unsigned long long* state;
  const float* ids_src; int64_t* ids_dst; int ids_ld, n_ids;
  int pack_blocks, draw_blocks;
};
template <bool BF16>
__global__ __launch_bounds__(256) void hn_prologue_kernel(const HnPrologueDev a) {
  const int blk = (int)blockIdx.x;
  if (blk < a.pack_blocks) {
    hn_pack_block<BF16>(a.pack, blk);
  } else if (blk < a.pack_blocks + a.draw_blocks) {
    hn_random_section<true>(a.draws, a.state, blk - a.pack_blocks, a.draw_blocks, a.ride);
  } else {
    const int i = (blk - a.pack_blocks - a.draw_blocks) * 256 + (int)threadIdx.x;
    if (i < a.n_ids) a.ids_dst[i] = (int64_t)a.ids_src[(size_t)i * a.ids_ld];      // float -> long: truncation toward zero, as ATen
  }
}

extern "C" int hn_render_prologue(int mode, const HnPackJob* pack_jobs, int n_pack, const HnDraw* draws_host, int n_draws,
                                  uint64_t* state_dev, const HnPrologue* p, hnStream_t stream) {
  HnPrologueDev a = {};
  int pb = 0;
  int rc = hn_pack_table_fill(pack_jobs, n_pack, a.pack, pb);
  if (rc != 0) return rc;
  long long db = 0;
  rc = hn_draw_table_fill(draws_host, n_draws, a.draws, db);
  if (rc != 0) return rc;
  if (db > 0 && state_dev == nullptr) return -3;
  if (mode != HN_MODE_BF16 && mode != HN_MODE_BF16_S8 && mode != HN_MODE_F32) return -2;
  a.state = (unsigned long long*)state_dev;
  a.ride.draw = -1;
  int ib = 0;
  if (p != nullptr) {
    if (p->t_rand_draw >= 0) {
      if (p->t_rand_draw >= n_draws || p->n_rays <= 0 || p->n <= 0) return -2;
      if (draws_host[p->t_rand_draw].kind != 0 || draws_host[p->t_rand_draw].n != (int64_t)p->n_rays * p->n) return -2;
      if (p->lower == nullptr || p->upper == nullptr || p->z_out == nullptr) return -3;
      if (p->pts_out != nullptr && (p->origins == nullptr || p->dirs == nullptr)) return -3;
      a.ride.draw = p->t_rand_draw; a.ride.n_rays = p->n_rays; a.ride.n = p->n; a.ride.ray_ld = p->ray_ld;
      a.ride.per_ray_bounds = p->per_ray_bounds; a.ride.scale = p->scale;
      a.ride.origins = p->origins; a.ride.dirs = p->dirs; a.ride.lower = p->lower; a.ride.upper = p->upper;
      a.ride.z_out = p->z_out; a.ride.pts_out = p->pts_out;
    }
    if (p->n_ids > 0) {
      if (p->ids_src == nullptr || p->ids_dst == nullptr || p->ids_ld < 1) return -3;
      a.ids_src = p->ids_src; a.ids_dst = p->ids_dst; a.ids_ld = p->ids_ld; a.n_ids = p->n_ids;
      ib = (p->n_ids + 255) / 256;
    }
  }
  a.pack_blocks = pb;
  a.draw_blocks = (int)db;
  const long long total = (long long)pb + db + ib;
  if (total == 0) return 0;
  if (mode == HN_MODE_F32)
    hipLaunchKernelGGL(hn_prologue_kernel<false>, dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(hn_prologue_kernel<true>, dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream, a);
  HN_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// loss head (losses.py:4-14): mean squared error of the coarse and the fine render against the same target
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void hn_mse_fwd_kernel(const float* __restrict__ c, const float* __restrict__ f,
                                                          const float* __restrict__ gt, long long n,
                                                          float* __restrict__ loss, float* __restrict__ d_c,
                                                          float* __restrict__ d_f) {
  __shared__ float part[2][16];
  float sc = 0.0f, sf = 0.0f;
  const float s1 = 1.0f * 2.0f / (float)n;      // hn_mse_bwd_kernel's factor for a root gradient of exactly 1
  for (long long i = threadIdx.x; i < n; i += 1024) {
    const float g = gt[i];
    const float dc = c[i] - g;
    sc += dc * dc;
    if (d_c != nullptr) d_c[i] = dc * s1;
    if (f != nullptr) {
      const float df = f[i] - g;
      sf += df * df;
      if (d_f != nullptr) d_f[i] = df * s1;
    }
  }
  sc = hn_wave_sum(sc);
  sf = hn_wave_sum(sf);
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { part[0][wv] = sc; part[1][wv] = sf; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0.0f, b = 0.0f;
    for (int k = 0; k < 16; ++k) { a += part[0][k]; b += part[1][k]; }
    // mean of each level first, then their sum: the reference adds two nn.MSELoss results
    loss[0] = a / (float)n + (f != nullptr ? b / (float)n : 0.0f);
  }
}
__global__ void hn_mse_bwd_kernel(const float* __restrict__ c, const float* __restrict__ f,
                                  const float* __restrict__ gt, long long n, const float* __restrict__ g_loss,
                                  float* __restrict__ dc, float* __restrict__ df) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float s = (g_loss != nullptr ? g_loss[0] : 1.0f) * 2.0f / (float)n;
  const float g = gt[i];
  dc[i] = (c[i] - g) * s;
  if (f != nullptr && df != nullptr) df[i] = (f[i] - g) * s;
}
extern "C" int hn_mse_loss_forward(const float* coarse, const float* fine, const float* gt, int64_t n, float* loss_out,
                                   hnStream_t stream) {
  if (n <= 0) return -2;
  if (coarse == nullptr || gt == nullptr || loss_out == nullptr) return -3;
  hipLaunchKernelGGL(hn_mse_fwd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, coarse, fine, gt, (long long)n,
                     loss_out, (float*)nullptr, (float*)nullptr);
  HN_CHECK_LAUNCH();
  return 0;
}
extern "C" int hn_mse_loss_forward_grad(const float* coarse, const float* fine, const float* gt, int64_t n,
                                        float* loss_out, float* d_coarse, float* d_fine, hnStream_t stream) {
  if (n <= 0) return -2;
  if (coarse == nullptr || gt == nullptr || loss_out == nullptr || d_coarse == nullptr) return -3;
  if (fine != nullptr && d_fine == nullptr) return -3;
  hipLaunchKernelGGL(hn_mse_fwd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, coarse, fine, gt, (long long)n,
                     loss_out, d_coarse, d_fine);
  HN_CHECK_LAUNCH();
  return 0;
}
extern "C" int hn_mse_loss_backward(const float* coarse, const float* fine, const float* gt, int64_t n,
                                    const float* g_loss, float* d_coarse, float* d_fine, hnStream_t stream) {
  if (n <= 0) return -2;
  if (coarse == nullptr || gt == nullptr || d_coarse == nullptr) return -3;
  if ((fine == nullptr) != (d_fine == nullptr)) return -3;
  hipLaunchKernelGGL(hn_mse_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, coarse,
                     fine, gt, (long long)n, g_loss, d_coarse, d_fine);
  HN_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// SE(3) exponential-map warp (warping.SE3Field.warp, hypernerf/warping.py:226-238; rigid_body.py:24-83)
//   theta = |w| ; a = w/theta ; b = v/theta
//   R = I + sin(theta) [a] + (1 - cos(theta)) [a]^2            (Modern Robotics 3.51)
//   t = (theta I + (1 - cos(theta)) [a] + (theta - sin(theta)) [a]^2) b      (3.88)
//   y = R p + t
// one thread per point; fp32, the same operation order in forward and backward
// ------------------------------------------------------------------------------------------------
struct HnV3 { float x, y, z; };
HN_DEV HnV3 hn_v3(float x, float y, float z) { HnV3 r = {x, y, z}; return r; }
HN_DEV HnV3 hn_cross(HnV3 a, HnV3 b) { return hn_v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
HN_DEV float hn_dot(HnV3 a, HnV3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
HN_DEV HnV3 hn_add(HnV3 a, HnV3 b) { return hn_v3(a.x + b.x, a.y + b.y, a.z + b.z); }
HN_DEV HnV3 hn_scale(float s, HnV3 a) { return hn_v3(s * a.x, s * a.y, s * a.z); }
HN_DEV HnV3 hn_load3(const float* p, size_t i) { return hn_v3(p[3 * i], p[3 * i + 1], p[3 * i + 2]); }
HN_DEV void hn_store3(float* p, size_t i, HnV3 v) { p[3 * i] = v.x; p[3 * i + 1] = v.y; p[3 * i + 2] = v.z; }

HN_DEV HnV3 hn_load3s(const float* p, size_t i, int ld) { return hn_v3(p[i * ld], p[i * ld + 1], p[i * ld + 2]); }
HN_DEV void hn_store3s(float* p, size_t i, int ld, HnV3 v) { p[i * ld] = v.x; p[i * ld + 1] = v.y; p[i * ld + 2] = v.z; }

// every operand with its own row stride (w and v are the two halves of the field's (P, 6) head output: no slicing
// copies); `rows_out` (optional, (P, 3 + H) with stride rows_ld) = [y | table[idx[ray]]]: the `warped_points` tensor of
// an axis-aligned-plane level (models.py:533-534, 578) written by the same launch — no index_select, no cat
__global__ void hn_se3_forward_kernel(const float* w, int w_ld, const float* v, int v_ld, const float* pts, int p_ld,
                                      int n, float* out, float* rows_out, int rows_ld, const float* table,
                                      const int64_t* idx, int H, int n_rows, int spr) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const HnV3 wv = hn_load3s(w, i, w_ld), vv = hn_load3s(v, i, v_ld), p = hn_load3s(pts, i, p_ld);
  const float theta = sqrtf(hn_dot(wv, wv));
  const HnV3 a = hn_scale(1.0f / theta, wv), b = hn_scale(1.0f / theta, vv);
  const float s = sinf(theta), c = 1.0f - cosf(theta), d = theta - s;
  const HnV3 q = hn_cross(a, p), r = hn_cross(a, q), m = hn_cross(a, b), nn = hn_cross(a, m);
  HnV3 y = hn_add(p, hn_add(hn_scale(s, q), hn_scale(c, r)));
  y = hn_add(y, hn_add(hn_scale(theta, b), hn_add(hn_scale(c, m), hn_scale(d, nn))));
  if (out != nullptr) hn_store3(out, i, y);
  if (rows_out != nullptr) {
    float* dst = rows_out + (size_t)i * rows_ld;
    dst[0] = y.x; dst[1] = y.y; dst[2] = y.z;
    if (table != nullptr && H > 0) {
      const long long row = idx[i / spr];
      const bool ok = row >= 0 && row < n_rows;      // as the machine's own gather: an index outside the table is NaN
      for (int k = 0; k < H; ++k) dst[3 + k] = ok ? table[(size_t)row * H + k] : __builtin_nanf("");
    }
  }
}

__global__ void hn_se3_backward_kernel(const float* w, int w_ld, const float* v, int v_ld, const float* pts, int p_ld,
                                       const float* gout, int g_ld, int n, float* dw, int dw_ld, float* dv, int dv_ld,
                                       float* dp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const HnV3 wv = hn_load3s(w, i, w_ld), vv = hn_load3s(v, i, v_ld), p = hn_load3s(pts, i, p_ld);
  const HnV3 g = hn_load3s(gout, i, g_ld);
  const float theta = sqrtf(hn_dot(wv, wv)), inv = 1.0f / theta;
  const HnV3 a = hn_scale(inv, wv), b = hn_scale(inv, vv);
  const float sn = sinf(theta), cs = cosf(theta), c = 1.0f - cs, d = theta - sn;
  const HnV3 q = hn_cross(a, p), r = hn_cross(a, q), m = hn_cross(a, b), nn = hn_cross(a, m);
  const HnV3 ga = hn_cross(g, a);                  // g x a
  const HnV3 gaa = hn_cross(ga, a);                // (g x a) x a
  // y = p + s q + c r + theta b + c m + d n
  const HnV3 g_p = hn_add(g, hn_add(hn_scale(sn, ga), hn_scale(c, gaa)));
  const HnV3 g_b = hn_add(hn_scale(theta, g), hn_add(hn_scale(c, ga), hn_scale(d, gaa)));
  HnV3 g_a = hn_scale(sn, hn_cross(p, g));
  g_a = hn_add(g_a, hn_scale(c, hn_add(hn_cross(q, g), hn_cross(p, ga))));
  g_a = hn_add(g_a, hn_scale(c, hn_cross(b, g)));
  g_a = hn_add(g_a, hn_scale(d, hn_add(hn_cross(m, g), hn_cross(b, ga))));
  // ds = cos, dc = sin, dd = 1 - cos = c ; the explicit theta of `theta b`
  float g_theta = cs * hn_dot(q, g) + sn * hn_dot(r, g) + hn_dot(b, g) + sn * hn_dot(m, g) + c * hn_dot(nn, g);
  // a = w / theta, b = v / theta
  g_theta -= inv * (hn_dot(g_a, a) + hn_dot(g_b, b));
  const HnV3 g_w = hn_add(hn_scale(inv, g_a), hn_scale(g_theta, a));   // d theta / d w = a
  const HnV3 g_v = hn_scale(inv, g_b);
  if (dw != nullptr) hn_store3s(dw, i, dw_ld, g_w);
  if (dv != nullptr) hn_store3s(dv, i, dv_ld, g_v);
  if (dp != nullptr) hn_store3(dp, i, g_p);
}

extern "C" int hn_se3_apply_forward(const float* w, const float* v, const float* points, int n_points, float* out,
                                    hnStream_t stream) {
  if (n_points <= 0) return -2;
  if (w == nullptr || v == nullptr || points == nullptr || out == nullptr) return -3;
  hipLaunchKernelGGL(hn_se3_forward_kernel, dim3((n_points + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, 3, v, 3,
                     points, 3, n_points, out, (float*)nullptr, 0, (const float*)nullptr, (const int64_t*)nullptr, 0, 0, 1);
  HN_CHECK_LAUNCH();
  return 0;
}

extern "C" int hn_se3_apply_backward(const float* w, const float* v, const float* points, const float* g_out,
                                     int n_points, float* d_w, float* d_v, float* d_points, hnStream_t stream) {
  if (n_points <= 0) return -2;
  if (w == nullptr || v == nullptr || points == nullptr || g_out == nullptr) return -3;
  hipLaunchKernelGGL(hn_se3_backward_kernel, dim3((n_points + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, 3, v, 3,
                     points, 3, g_out, 3, n_points, d_w, 3, d_v, 3, d_points);
  HN_CHECK_LAUNCH();
  return 0;
}

extern "C" int hn_se3_warp_forward(const float* w, int w_ld, const float* v, int v_ld, const float* points, int p_ld,
                                   int n_points, float* out, float* rows_out, int rows_ld, const float* table,
                                   const int64_t* idx, int H, int n_rows, int samples_per_ray, hnStream_t stream) {
  if (n_points <= 0 || w_ld < 3 || v_ld < 3 || p_ld < 3) return -2;
  if (w == nullptr || v == nullptr || points == nullptr || (out == nullptr && rows_out == nullptr)) return -3;
  if (rows_out != nullptr) {
    if (H < 0 || rows_ld < 3 + H || samples_per_ray <= 0) return -2;
    if (H > 0 && (table == nullptr || idx == nullptr || n_rows <= 0)) return -3;
  }
  hipLaunchKernelGGL(hn_se3_forward_kernel, dim3((n_points + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, w_ld, v,
                     v_ld, points, p_ld, n_points, out, rows_out, rows_ld, table, idx, H, n_rows,
                     samples_per_ray > 0 ? samples_per_ray : 1);
  HN_CHECK_LAUNCH();
  return 0;
}

extern "C" int hn_se3_warp_backward(const float* w, int w_ld, const float* v, int v_ld, const float* points, int p_ld,
                                    const float* g_out, int g_ld, int n_points, float* d_w, int dw_ld, float* d_v,
                                    int dv_ld, float* d_points, hnStream_t stream) {
  if (n_points <= 0 || w_ld < 3 || v_ld < 3 || p_ld < 3 || g_ld < 3) return -2;
  if (w == nullptr || v == nullptr || points == nullptr || g_out == nullptr) return -3;
  if ((d_w != nullptr && dw_ld < 3) || (d_v != nullptr && dv_ld < 3)) return -2;
  hipLaunchKernelGGL(hn_se3_backward_kernel, dim3((n_points + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, w_ld, v,
                     v_ld, points, p_ld, g_out, g_ld, n_points, d_w, dw_ld, d_v, dv_ld, d_points);
  HN_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// hardware layout probe (run once on a real MI355X by tests/test_gpu_probe.py)
// ------------------------------------------------------------------------------------------------
__global__ void hn_probe_kernel(float* out_bf16, float* out_f32, float* out_glds, const float* src) {
  __shared__ __attribute__((aligned(16))) char smem[1024];
  const int lane = threadIdx.x;
  const int r = lane & 31, h = lane >> 5;
  // probe 1: A[i][k] = i, B[k][j] = (k == j % 16)            -> D[i][j] = i
  // probe 2: A[i][k] = k (k = 8h+j as documented), same B     -> D[i][j] = j % 16
  // Together: A row = lane&31, B col = lane&31, element (h,j) of A meets element (h,j) of B, C/D map.
  bf16x8 a1, a2, b;
  for (int j = 0; j < 8; ++j) {
    const int k = 8 * h + j;
    a1[j] = (__bf16)(float)r;
    a2[j] = (__bf16)(float)k;
    b[j] = (__bf16)((k == (r & 15)) ? 1.0f : 0.0f);
  }
  f32x16 c = {0}, c2 = {0};
  c = hn_mfma_bf16(a1, b, c);
  c2 = hn_mfma_bf16(a2, b, c2);
  for (int i = 0; i < 16; ++i) out_bf16[lane * 16 + i] = c[i];
  for (int i = 0; i < 16; ++i) out_bf16[1024 + lane * 16 + i] = c2[i];
  // fp32: A[i][k] = i + 1000*k (k = h), B[k][j] = (k==0 ? 1 : 0.5*j)  -> D[i][j] = i + (i+1000)*0.5*j
  f32x16 d = {0};
  d = hn_mfma_f32((float)(r + 1000 * h), h == 0 ? 1.0f : 0.5f * (float)r, d);
  for (int i = 0; i < 16; ++i) out_f32[lane * 16 + i] = d[i];
  // LDS-DMA: 16 B per lane from src into LDS, read back
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + lane * 4),
                                   (__attribute__((address_space(3))) void*)smem, 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = 0; i < 4; ++i) out_glds[lane * 4 + i] = reinterpret_cast<float*>(smem)[lane * 4 + i];
}

extern "C" int hn_probe_mfma(float* out_bf16_acc, float* out_f32_acc, float* out_glds, hnStream_t stream) {
  if (out_bf16_acc == nullptr || out_f32_acc == nullptr || out_glds == nullptr) return -3;
  // source for the LDS-DMA test: reuse out_glds's second half (256 floats) — caller fills [256,512) with a ramp
  hipLaunchKernelGGL(hn_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out_bf16_acc, out_f32_acc, out_glds,
                     out_glds + 256);
  HN_CHECK_LAUNCH();
  return 0;
}
