// vmvm_probe_hooks.h (probe builds): the instrumented twin of pytorch_empirical_mvm_amd/csrc/hooks/vmvm_probe_hooks.h.  Put this
// directory in front of csrc on the include path (-I tools/probe/hooks -I pytorch_empirical_mvm_amd/csrc) and choose the experiment
// with -D switches -- all of them live HERE, the kernels only call the hooks:
//   -DVMVM_PROBE_EPI=1|2|3        128x128 persistent GEMM without global stores / without epilogue math / neither (epi_ablation.sh)
//   -DVMVM_PROBE_GM=n  -DVMVM_PROBE_GM_PP=n     M panels per raster group (raster_probe.sh)
//   -DVMVM_PROBE_ONE_WG           one workgroup of the persistent GEMM per CU (one_wg_probe.sh)
//   -DVMVM_PROBE_TIMELINE         100 MHz stamps of (main loop start, epilogue start, epilogue end) per tile, per CU (timeline_probe.sh)
//   -DVMVM_PROBE_STAGGER=P / -DVMVM_PROBE_STAGGER_CU=pct     start-up staggers (stagger128_probe.sh, stagger_cu_probe.sh)
//   -DVMVM_PROBE_BUILD            LayerNorm backward: workgroups per CU from VMVM_LN_PER_CU (tools/gpu_check.py benchln)
//   -DW3_TIMELINE                 cycle stamps of waves 0 and 5 of attn_bwd_dkv_win3_kernel (tools/scratch/w3_timeline.py)
//   -DW4_TIMELINE                 s_memtime stamps of every wave of workgroup 0, sequences 2..5, of the win4 kernels into the buffer passed as
//                                 vmvm_attn_fwd_desc.drop_mask ([4 sequences][13 waves][32 stamps] u64; tools/scratch/w4_timeline.py)
//   -DW4_NO_ODD                   win4 forward: wave 12 runs the two-tile walk like the others (cost of the odd tile)
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

namespace vmvm_hook {

#ifndef VMVM_PROBE_EPI
#define VMVM_PROBE_EPI 0
#endif
#ifndef VMVM_PROBE_GM
#define VMVM_PROBE_GM 8
#endif
constexpr int EPI = VMVM_PROBE_EPI;
constexpr int GM = VMVM_PROBE_GM;
#ifdef VMVM_PROBE_GM_PP
template <int WM> constexpr int GM_PP = VMVM_PROBE_GM_PP;
#else
template <int WM> constexpr int GM_PP = 16 / WM;
#endif
#ifdef VMVM_PROBE_ONE_WG
constexpr bool ONE_WG = true;
#else
constexpr bool ONE_WG = false;
#endif

#ifdef VMVM_PROBE_TIMELINE
constexpr int TL_TILES = 24;
__device__ unsigned g_tl_arrivals[8 * 256];
__device__ unsigned long long g_tl[8 * 256][2][TL_TILES][4];
struct GemmTimeline {
  unsigned key, slot; int t;
  __device__ __forceinline__ void init(unsigned char* smem, int tid) {
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    key = (blockIdx.x & 7) * 256 + ((hw >> 8) & 0xff);
    t = 0;
    if (tid == 0) *reinterpret_cast<unsigned*>(smem) = atomicAdd(&g_tl_arrivals[key], 1u);
    __syncthreads();
    slot = *reinterpret_cast<volatile unsigned*>(smem) & 1;
    __syncthreads();
  }
  __device__ __forceinline__ void stamp(int which, int tid) {
    if (tid == 0 && t < TL_TILES) {
      g_tl[key][slot][t][which] = wall_clock64();
      if (which == 0) g_tl[key][slot][t][3] = clock64();
    }
  }
  __device__ __forceinline__ void next_tile() { ++t; }
};
static void probe_timeline_reset() {
  static unsigned long long z[8 * 256 * 2 * TL_TILES * 4];
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tl), z, sizeof(z));
}
static void probe_timeline_dump(int n_cus) {
  static unsigned long long h[8 * 256][2][TL_TILES][4];
  (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_tl), sizeof(h));
  int shown = 0;
  for (int key = 0; key < 8 * 256 && shown < n_cus; ++key) {
    if (!h[key][0][0][0] || !h[key][1][0][0]) continue;
    const unsigned long long t0 = h[key][0][0][0] < h[key][1][0][0] ? h[key][0][0][0] : h[key][1][0][0];
    printf("CU key %d (xcd %d, hw %d): times in us from the first workgroup's start; per tile: main start | epilogue start | epilogue end\n", key, key >> 8, key & 255);
    for (int sl = 0; sl < 2; ++sl)
      if (h[key][sl][TL_TILES - 1][0]) printf("   wg%d shader clock over tiles 0..%d: %.0f MHz\n", sl, TL_TILES - 1,
                                              (double)(h[key][sl][TL_TILES - 1][3] - h[key][sl][0][3]) / ((h[key][sl][TL_TILES - 1][0] - h[key][sl][0][0]) * 0.01));
    for (int t = 0; t < TL_TILES; ++t) {
      for (int sl = 0; sl < 2; ++sl) {
        if (!h[key][sl][t][0]) { printf("   wg%d t%-2d  -                          ", sl, t); continue; }
        printf("   wg%d t%-2d %7.2f %7.2f %7.2f (main %5.2f epi %5.2f)", sl, t, (h[key][sl][t][0] - t0) * 0.01, (h[key][sl][t][1] - t0) * 0.01, (h[key][sl][t][2] - t0) * 0.01,
               (h[key][sl][t][1] - h[key][sl][t][0]) * 0.01, (h[key][sl][t][2] - h[key][sl][t][1]) * 0.01);
      }
      printf("\n");
    }
    ++shown;
  }
}
#else
struct GemmTimeline {
  __device__ __forceinline__ void init(unsigned char*, int) {}
  __device__ __forceinline__ void stamp(int, int) {}
  __device__ __forceinline__ void next_tile() {}
};
#endif

#ifdef VMVM_PROBE_STAGGER_CU
__device__ unsigned g_probe_cu_arrivals[8 * 256];
#endif
// nk_tile = K tiles of the workgroup's first tile (its estimated duration in 64-cycle units: 17 per K tile + 100)
__device__ __forceinline__ void gemm_stagger(unsigned char* smem, int tid, int li, int nk_tile) {
#ifdef VMVM_PROBE_STAGGER_CU   /* the SECOND workgroup to arrive on a compute unit (arrival order per CU through HW_ID) starts a fraction of a tile time late */
  {
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    const unsigned key = (blockIdx.x & 7) * 256 + ((hw >> 8) & 0xff);
    if (tid == 0) *reinterpret_cast<unsigned*>(smem) = atomicAdd(&g_probe_cu_arrivals[key], 1u);
    __syncthreads();
    const unsigned old = *reinterpret_cast<volatile unsigned*>(smem);
    __syncthreads();
    const int tile64 = nk_tile * 17 + 100;
    const int wait64 = (old & 1) ? tile64 * VMVM_PROBE_STAGGER_CU / 100 : 0;
    for (int t = 0; t < wait64; t += 100) __builtin_amdgcn_s_sleep(100);
  }
#endif
#ifdef VMVM_PROBE_STAGGER      /* workgroup i of an XCD starts (i % P) / P of an estimated tile time late (are the two workgroups of a CU in lockstep?) */
  {
    const int tile64 = nk_tile * 17 + 100;
    const int wait64 = (li % VMVM_PROBE_STAGGER) * tile64 / VMVM_PROBE_STAGGER;
    for (int t = 0; t < wait64; t += 100) __builtin_amdgcn_s_sleep(100);
  }
#endif
  (void)smem; (void)tid; (void)li; (void)nk_tile;
}

inline int ln_bwd_per_cu(int per_cu) {
#ifdef VMVM_PROBE_BUILD
  if (const char* e = getenv("VMVM_LN_PER_CU")) { const int v = atoi(e); if (v >= 1 && v <= 8) per_cu = v; }
#endif
  return per_cu;
}

#ifdef W3_TIMELINE
__device__ unsigned long long w3_dbg[8192];
struct W3Timeline {
  bool on; unsigned long long* buf; int n;
  __device__ __forceinline__ W3Timeline(int b, int lane, int wave) : on(blockIdx.x == 8 && b == 2 && lane == 0 && (wave == 0 || wave == 5)), buf(w3_dbg + (wave == 0 ? 0 : 2048)), n(1) {}
  __device__ __forceinline__ void stamp(int slot) { if (on) buf[n++] = ((unsigned long long)slot << 48) | (__builtin_readcyclecounter() & 0xffffffffffffull); }
  __device__ __forceinline__ void flush(int) { if (on) buf[0] = n; }
};
extern "C" inline int vmvm_w3_debug_read(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(w3_dbg), (size_t)n * 8, 0, hipMemcpyDeviceToHost);
}
#else
struct W3Timeline {
  __device__ __forceinline__ W3Timeline(int, int, int) {}
  __device__ __forceinline__ void stamp(int) {}
  __device__ __forceinline__ void flush(int) {}
};
#endif

#ifdef W4_TIMELINE
constexpr bool W4_TIMELINE_BUILD = true;
__device__ __forceinline__ void w4_stamp(void* buf, int block, int b, int lane, int wave, int idx) {
  if (buf && block == 0 && b >= 2 && b < 6 && lane == 0) reinterpret_cast<unsigned long long*>(buf)[((b - 2) * 13 + wave) * 32 + idx] = __builtin_readcyclecounter();
}
#else
constexpr bool W4_TIMELINE_BUILD = false;
__device__ __forceinline__ void w4_stamp(void*, int, int, int, int, int) {}
#endif
#ifdef W4_NO_ODD
constexpr bool W4_SKIP_ODD = true;
#else
constexpr bool W4_SKIP_ODD = false;
#endif

}  // namespace vmvm_hook
#ifdef VMVM_PROBE_TIMELINE
using vmvm_hook::probe_timeline_reset;
using vmvm_hook::probe_timeline_dump;
#endif
