// attn_win3.h -- compile-time geometry of the win_layout = 1 window-attention kernels (attention_win3.hip): window (8,7,7),
// un-shifted or shifted by (0,3,3) (video_swin.py:95-108,292-307), tokens of a window in the order of swin_index.win3_perm:
//   slot = tile * 16 + l * 8 + d        tile 0..24, l = position inside the tile's pair of (h, w) positions, d = temporal slice.
// The 49 (h, w) positions come region-major -- classes A (h<4,w<4) tiles 0-7, B (h<4,w>=4) 8-13, C (h>=4,w<4) 14-19, D (h>=4,w>=4)
// 20-24 (tile 24 holds ONE position: tokens 392..399 are padding) -- so every tile lies in one mask region of every window type and a
// (query tile, key tile) pair is fully live or fully masked.  Window type bits: 1 = the window is split along h (regions {A,B} |
// {C,D}), 2 = split along w ({A,C} | {B,D}); class bits: bit 1 = h half, bit 0 = w half.
// The relative-position index is (dq - dk + 7) * 169 + rho, rho = A(q) - A(k) + 84 with A = 13 h + w (video_swin.py:123-137): linear
// in A, so a lane's table row is a lane constant plus a per-tile immediate; the two positions of a tile are w-neighbours (A step 1)
// or h-neighbours (A step 13).
#pragma once

namespace w3 {

constexpr int NT = 25;                       // 16-token tiles per window
constexpr int L = 392;
constexpr int ROWB = 256;                    // bytes of one row of the windowed table: 12 windows of 4 consecutive f32 entries (192 bytes) padded to 64 banks --
                                             // at 192 the two rows a b128 read touches sit 48 banks apart and (row + 1, window s) collides with (row, s - 4): 18-34 % of
                                             // the LDS cycles of the round-4 backward kernels were bank conflicts (profiles/r04_pmc_window_attention_*.txt, r05_pmc_*)
constexpr int NROW = 170;                    // 169 rows rho + the padding row (-inf)
constexpr int TAB_BYTES = NROW * ROWB;       // 43 520
constexpr int PH[50] = {0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3,            // A
                        0, 1, 2, 3, 0, 1, 2, 3, 0, 1, 2, 3,                        // B (pairs along h)
                        4, 4, 4, 4, 5, 5, 5, 5, 6, 6, 6, 6,                        // C
                        4, 4, 5, 5, 6, 6, 4, 5, 6, 6};                             // D (+ the padding slot, any valid position)
constexpr int PW[50] = {0, 1, 2, 3, 0, 1, 2, 3, 0, 1, 2, 3, 0, 1, 2, 3,
                        4, 4, 4, 4, 5, 5, 5, 5, 6, 6, 6, 6,
                        0, 1, 2, 3, 0, 1, 2, 3, 0, 1, 2, 3,
                        4, 5, 4, 5, 4, 5, 6, 6, 6, 6};
constexpr int posA(int i) { return 13 * PH[i] + PW[i]; }
constexpr int tileA0(int t) { return posA(2 * t); }                                  // A of the tile's first position
constexpr int tileStep(int t) { return t == NT - 1 ? 1 : posA(2 * t + 1) - posA(2 * t); }   // 1 or 13
constexpr int CB[5] = {0, 8, 14, 20, 25};                                            // first tile of each class
constexpr int cls_of(int t) { return t < 8 ? 0 : t < 14 ? 1 : t < 20 ? 2 : 3; }
// classes a query / key of class c attends to under window type wt
constexpr int live_mask(int c, int wt) {
  int m = 0;
  for (int k = 0; k < 4; ++k) {
    const bool ok = (!(wt & 1) || ((c >> 1) == (k >> 1))) && (!(wt & 2) || ((c & 1) == (k & 1)));
    if (ok) m |= 1 << k;
  }
  return m;
}

struct TileList { int n; int t[NT]; };
// all tiles of the classes in mask m4, class order
constexpr TileList list_all(int m4) {
  TileList l{};
  for (int c = 0; c < 4; ++c)
    if (m4 & (1 << c))
      for (int t = CB[c]; t < CB[c + 1]; ++t) l.t[l.n++] = t;
  return l;
}
// the part `part` of `nparts` of every class in m4 (each class is cut into nparts runs whose sizes differ by at most one; which
// parts get a class's longer runs rotates (ROT) so the parts come out even: 4 parts = 7 / 6 / 6 / 6 tiles, 3 parts = 8 / 9 / 8, 2 parts = 13 / 12)
constexpr int ROT[4] = {0, 0, 2, 4};             // (used modulo nparts: class D rotates by one only for three parts -> 8 / 9 / 8 tiles)
constexpr int part_len(int c, int part, int nparts) {
  const int n = CB[c + 1] - CB[c], q = n / nparts, r = n % nparts;
  return q + ((((part + nparts - ROT[c] % nparts) % nparts) < r) ? 1 : 0);
}
constexpr int part_beg(int c, int part, int nparts) {
  int b = CB[c];
  for (int p = 0; p < part; ++p) b += part_len(c, p, nparts);
  return b;
}
constexpr TileList list_part(int m4, int part, int nparts) {
  TileList l{};
  for (int c = 0; c < 4; ++c)
    if (m4 & (1 << c))
      for (int i = 0; i < part_len(c, part, nparts); ++i) l.t[l.n++] = part_beg(c, part, nparts) + i;
  return l;
}
// index of tile t inside the part's full (mask 15) list: where its running sums live
constexpr int max_part(int nparts) {                 // tiles of the longest part (mask 15)
  int m = 0;
  for (int p = 0; p < nparts; ++p) { const int n = list_part(15, p, nparts).n; m = n > m ? n : m; }
  return m;
}
constexpr int slot_in_part(int t, int part, int nparts) {
  const TileList l = list_part(15, part, nparts);
  for (int i = 0; i < l.n; ++i)
    if (l.t[i] == t) return i;
  return -1;
}

}  // namespace w3
