// attn_win3_dev.h -- device helpers shared by the win_layout = 1 window-attention translation units (attention_win3.hip: the
// 16 x 392 score-block kernels of round 4; attention_win4.hip: the key-blocked kernels of round 5): the windowed copy of a head's
// relative-position table, position codes, window types / live classes, paired transposing reads.
#pragma once
#include "attn_common.h"
#include "attn_win3.h"

namespace {

// windowed copy of one head's table column (stage: the 2535 entries [delta * 169 + rho] in LDS): row rho holds the 12 windows of 4
// consecutive entries a lane can need.  DIR 0 (a lane's 4 values are consecutive KEYS, delta falls): entry j of window s is
// delta = 14 - s - j; DIR 1 (consecutive QUERIES, delta rises): delta = s + j.  Row 169 = -inf (padding tokens).
template <int DIR>
__device__ __forceinline__ void w3_build_table(unsigned char* tl, const float* stage, int tid, int nthreads) {
  for (int i = tid; i < w3::NROW * 12; i += nthreads) {
    const int rho = i / 12, s = i - rho * 12;
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e = DIR ? s + j : 14 - s - j;
      v[j] = rho < 169 ? stage[e * 169 + rho] : NEG_INF;
    }
    *reinterpret_cast<f32x4*>(tl + rho * w3::ROWB + s * 16) = v;
  }
}
// A = 13 h + w of slot position `pos` (run-time index, set-up code only): a namespace-scope table (a function-local constexpr array
// indexed at run time gets copied to scratch per lane)
__device__ const int W3_PA[50] = {w3::posA(0),  w3::posA(1),  w3::posA(2),  w3::posA(3),  w3::posA(4),  w3::posA(5),  w3::posA(6),  w3::posA(7),  w3::posA(8),  w3::posA(9),
                                  w3::posA(10), w3::posA(11), w3::posA(12), w3::posA(13), w3::posA(14), w3::posA(15), w3::posA(16), w3::posA(17), w3::posA(18), w3::posA(19),
                                  w3::posA(20), w3::posA(21), w3::posA(22), w3::posA(23), w3::posA(24), w3::posA(25), w3::posA(26), w3::posA(27), w3::posA(28), w3::posA(29),
                                  w3::posA(30), w3::posA(31), w3::posA(32), w3::posA(33), w3::posA(34), w3::posA(35), w3::posA(36), w3::posA(37), w3::posA(38), w3::posA(39),
                                  w3::posA(40), w3::posA(41), w3::posA(42), w3::posA(43), w3::posA(44), w3::posA(45), w3::posA(46), w3::posA(47), w3::posA(48), w3::posA(49)};
__device__ __forceinline__ int w3_posA_rt(int pos) { return W3_PA[pos]; }
// window type of window position w from its (tile-uniform) region row: bit 0 = split along h (A | C differ), bit 1 = along w (A | B)
__device__ __forceinline__ int w3_window_type(const uint8_t* region, int w) {
  const uint8_t* rw = region + (size_t)w * w3::L;
  const int ra = rw[0], rb = rw[w3::CB[1] * 16], rcl = rw[w3::CB[2] * 16];
  return (ra != rcl ? 1 : 0) | (ra != rb ? 2 : 0);
}
__device__ __forceinline__ int w3_live_rt(int c, int wt) {
  int m = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const bool ok = (!(wt & 1) || ((c >> 1) == (k >> 1))) && (!(wt & 2) || ((c & 1) == (k & 1)));
    m |= ok ? (1 << k) : 0;
  }
  return m;
}
// transposing reads of TWO tiles (byte offsets offa / offb from the lane bases pa: hd 0-15, pc: hd 16-31), asm as tr_read4
__device__ __forceinline__ void tr_read4_2(s16x4& a0, s16x4& a1, s16x4& c0, s16x4& c1, uint32_t pa, uint32_t pc, const int offa, const int offb) {
  asm volatile("ds_read_b64_tr_b16 %0, %4 offset:%6\n\tds_read_b64_tr_b16 %1, %4 offset:%7\n\t"
               "ds_read_b64_tr_b16 %2, %5 offset:%6\n\tds_read_b64_tr_b16 %3, %5 offset:%7"
               : "=&v"(a0), "=&v"(a1), "=&v"(c0), "=&v"(c1) : "v"(pa), "v"(pc), "i"(offa), "i"(offb) : "memory");
}


}  // namespace
