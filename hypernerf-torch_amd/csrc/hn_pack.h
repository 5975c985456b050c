// Weight packing, shared by hn_mlp.hip (hn_pack_units / hn_pack_units_multi) and hn_render.hip (hn_render_prologue, which
// packs the step's weight streams in the same launch that draws its random numbers and places its coarse samples).
#pragma once
#include "hn_common.h"

template <bool BF16>
HN_DEV void hn_pack_one(const HnPackUnit* units, int n_units, const float* const* ptrs, char* out,
                        const HnPackBias* bias, int n_bias, float* bias_out, int wid) {
  const int lane = threadIdx.x & 63;
  const int row = lane & 31, h = lane >> 5;
  if (wid < n_units) {
    const HnPackUnit u = units[wid];
    const float* W = u.w_id >= 0 ? ptrs[u.w_id] : nullptr;
    auto fetch = [&](int k) -> float {
      int sr, sc;
      if (u.transposed) { sr = u.r0 + k; sc = u.c0 + row; }
      else { sr = u.r0 + row; sc = u.c0 + k; }
      if (W == nullptr || sr < 0 || sc < 0 || sr >= u.r_end || sc >= u.c_end) return 0.0f;
      return W[(size_t)sr * u.ld + sc];
    };
    char* dst = out + (size_t)wid * 1024 + lane * 16;
    if (BF16) {
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (__bf16)fetch(u.k0 + hn_pi16(h, j));
      *reinterpret_cast<bf16x8*>(dst) = o;
    } else {
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = fetch(hn_rho(u.k0 + e, h));
      *reinterpret_cast<f32x4*>(dst) = o;
    }
  } else if (wid - n_units < n_bias) {
    const HnPackBias b = bias[wid - n_units];
    const float* src = b.w_id >= 0 ? ptrs[b.w_id] : nullptr;
    for (int i = lane; i < b.len; i += 64) bias_out[b.off + i] = (src != nullptr && i < b.n) ? src[i] : 0.0f;
  }
}
// several programs' streams in ONE launch (a training step packs three: each launch costs more in dispatch than in work)
struct HnPackTable {
  HnPackJob j[HN_MAX_PACK_JOBS];
  int first_block[HN_MAX_PACK_JOBS + 1];
  int n;
};
// Two units per wave, their load chains side by side (round 6): a unit is three dependent loads deep — descriptor, source
// pointer, the 8 weights of the lane — and the ~10,000 waves of a step's pack fill the chip only once, so the launch is
// one chain long; with two units per wave the chains overlap and the grid halves.  Every load is made unconditionally
// from a clamped address and the out-of-range value replaced afterwards (no branch between the loads).
template <bool BF16>
HN_DEV void hn_pack_pair(const HnPackUnit* __restrict__ units, int n_units, const float* const* __restrict__ ptrs,
                         char* __restrict__ out, int u0) {
  const int lane = threadIdx.x & 63;
  const int row = lane & 31, h = lane >> 5;
  constexpr int NE = BF16 ? 8 : 4;
  HnPackUnit u[2];
  const float* W[2];
  float v[2][NE];
  bool ok[2][NE];
#pragma unroll
  for (int q = 0; q < 2; ++q) u[q] = units[min(u0 + q, n_units - 1)];
#pragma unroll
  for (int q = 0; q < 2; ++q) W[q] = u[q].w_id >= 0 ? ptrs[u[q].w_id] : nullptr;
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int k = BF16 ? u[q].k0 + hn_pi16(h, e) : hn_rho(u[q].k0 + e, h);
      int sr, sc;
      if (u[q].transposed) { sr = u[q].r0 + k; sc = u[q].c0 + row; }
      else { sr = u[q].r0 + row; sc = u[q].c0 + k; }
      ok[q][e] = W[q] != nullptr && sr >= 0 && sc >= 0 && sr < u[q].r_end && sc < u[q].c_end;
      v[q][e] = ok[q][e] ? W[q][(size_t)sr * u[q].ld + sc] : 0.0f;
    }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    if (u0 + q >= n_units) continue;
    char* dst = out + (size_t)(u0 + q) * 1024 + lane * 16;
    if constexpr (BF16) {
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (__bf16)v[q][e];
      *reinterpret_cast<bf16x8*>(dst) = o;
    } else {
      *reinterpret_cast<f32x4*>(dst) = f32x4{v[q][0], v[q][1], v[q][2], v[q][3]};
    }
  }
}
// block `blk` (0 .. first_block[n]) of a multi-program pack: 4 waves, two 1-KiB units (or one bias record) each
template <bool BF16>
HN_DEV void hn_pack_block(const HnPackTable& tab, int blk) {
  int k = 0;
#pragma unroll
  for (int i = 1; i < HN_MAX_PACK_JOBS; ++i)
    if (i < tab.n && blk >= tab.first_block[i]) k = i;
  const HnPackJob jb = tab.j[k];
  const int wv = (blk - tab.first_block[k]) * 4 + ((threadIdx.x >> 6) & 3);
  const int unit_waves = (jb.n_units + 1) / 2;
  if (wv < unit_waves) hn_pack_pair<BF16>(jb.units, jb.n_units, jb.ptrs, (char*)jb.wstream, 2 * wv);
  else hn_pack_one<BF16>(jb.units, 0, jb.ptrs, (char*)jb.wstream, jb.bias, jb.n_bias, jb.bias_out, wv - unit_waves);
}
// host: argument checks + block ranges of the jobs (256-thread blocks); returns 0 or a negative status
static inline int hn_pack_table_fill(const HnPackJob* jobs, int n_jobs, HnPackTable& tab, int& blocks) {
  if (n_jobs < 0 || n_jobs > HN_MAX_PACK_JOBS) return -1;
  blocks = 0;
  tab.n = 0;
  if (n_jobs == 0) return 0;
  if (jobs == nullptr) return -3;
  for (int i = 0; i < n_jobs; ++i) {
    if (jobs[i].n_units < 0 || jobs[i].n_bias < 0) return -1;
    if (jobs[i].n_units > 0 && (jobs[i].units == nullptr || jobs[i].ptrs == nullptr || jobs[i].wstream == nullptr)) return -3;
    if (jobs[i].n_bias > 0 && (jobs[i].bias == nullptr || jobs[i].bias_out == nullptr || jobs[i].ptrs == nullptr)) return -3;
    tab.j[i] = jobs[i];
    tab.first_block[i] = blocks;
    blocks += ((jobs[i].n_units + 1) / 2 + jobs[i].n_bias + 3) / 4;      // two units or one bias record per wave
  }
  tab.first_block[n_jobs] = blocks;
  tab.n = n_jobs;
  return 0;
}
