// gemm_probe.hip -- standalone harness for the GEMM main loops of libvmvm (no torch): correctness against a naive f32 reference
// on sampled rows + HIP-event timing of the 256x256 ping-pong kernel variants next to the 128x128 persistent kernel, on uniform
// random [-1,1) bf16 operands, interleaved rounds in ONE process.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-pass-failed -I tools/probe/hooks -I pytorch_empirical_mvm_amd/csrc tools/probe/gemm_probe.hip -o tools/probe/gemm_probe
//   tools/probe/gemm_probe [shape-set] [reps]
#include "../../pytorch_empirical_mvm_amd/csrc/gemm.hip"
#include "../../pytorch_empirical_mvm_amd/csrc/gemm_pp.hip"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

thread_local int g_vmvm_last_hip_error = 0;
extern "C" int vmvm_colsum_bf16(const void*, int32_t, int32_t, int32_t, const float*, int32_t, float*, int32_t, void*) { return VMVM_ENOSUPPORT; }
int vmvm_colsum_scaled(const void*, int32_t, int32_t, int32_t, float, float*, void*) { return VMVM_ENOSUPPORT; }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void fill_kernel(u16* p, size_t n, uint32_t seed) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    uint32_t x = (uint32_t)i * 2654435761u ^ seed;
    x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
    const float f = (float)(x >> 8) * (2.0f / 16777216.0f) - 1.0f;       // uniform [-1, 1)
    p[i] = f2bf(f);
  }
}
__global__ void fill_f32_kernel(float* p, size_t n, uint32_t seed) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { uint32_t x = (uint32_t)i * 2654435761u ^ seed; x ^= x >> 15; x *= 0x2c1b3c6du; x ^= x >> 12; p[i] = (float)(x >> 8) * (1.0f / 16777216.0f) - 0.5f; }
}
// reference rows: R[r][n] = sum_k A[rows[r]][k] * B[n][k]   (f32 accumulate)
__global__ void ref_rows_kernel(const u16* A, const u16* B, const int* rows, float* R, int N, int K, int lda, int ldb) {
  const int r = blockIdx.y, n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const u16* a = A + (size_t)rows[r] * lda;
  const u16* b = B + (size_t)n * ldb;
  float s = 0.f;
  for (int k = 0; k < K; ++k) s = fmaf(bf2f(a[k]), bf2f(b[k]), s);
  R[(size_t)r * N + n] = s;
}

// weight-gradient form: R[r][n] = sum_k A[k][rows[r]] * B[k][n]   (A [K][M], B [K][N])
__global__ void ref_rows_tn_kernel(const u16* A, const u16* B, const int* rows, float* R, int M, int N, int K) {
  const int r = blockIdx.y, n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const int m = rows[r];
  float s = 0.f;
  for (int k = 0; k < K; ++k) s = fmaf(bf2f(A[(size_t)k * M + m]), bf2f(B[(size_t)k * N + n]), s);
  R[(size_t)r * N + n] = s;
}

// main loop of the TN build alone: square problems, no split, (a) full, (b) no epilogue, (c) phase timers
static int run_tn_mainloop(hipStream_t st, int reps) {
  const int M = 4096, N = 4096, K = 8192;
  u16 *A, *B; float* C; unsigned* dbg;
  CK(hipMalloc(&A, (size_t)K * M * 2)); CK(hipMalloc(&B, (size_t)K * N * 2)); CK(hipMalloc(&C, (size_t)M * N * 4)); CK(hipMalloc(&dbg, 256));
  fill_kernel<<<2048, 256, 0, st>>>(A, (size_t)K * M, 0x1234u);
  fill_kernel<<<2048, 256, 0, st>>>(B, (size_t)K * N, 0x9876u);
  vmvm_gemm_desc d;
  memset(&d, 0, sizeof(d));
  d.A = A; d.B = B; d.C = C; d.M = M; d.N = N; d.K = K; d.lda = M; d.ldb = N; d.ldc = N; d.col_scale = 1.f; d.out_fp32 = 1; d.splitk = 1;
  struct V { const char* name; int (*fn)(const vmvm_gemm_desc&, hipStream_t); };
  const V vs[] = {{"tn64x2", launch_pp_f<false, false, (EF_SPLIT | EF_F32), 2, 64, 2>}, {"tn32x4", launch_pp_f<false, false, (EF_SPLIT | EF_F32), 2, 32, 4>},
                  {"tn64x2_ne", launch_pp_f<false, false, (EF_SPLIT | EF_F32), 2, 64, 2, 1>}, {"tn32x4_ne", launch_pp_f<false, false, (EF_SPLIT | EF_F32), 2, 32, 4, 1>},
                  {"tn64x2_t", launch_pp_f<false, false, (EF_SPLIT | EF_F32), 2, 64, 2, 2>},
                  {"nt64x2_ne(ref)", launch_pp_f<true, true, 0, 2, 64, 2, 1>}};
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("TN main loop M=%d N=%d K=%d\n", M, N, K);
  for (const V& v : vs) {
    vmvm_gemm_desc dv = d;
    if (strstr(v.name, "nt")) { dv.a_kmajor = dv.b_kmajor = 1; dv.lda = dv.ldb = K; dv.out_fp32 = 0; }
    if (strstr(v.name, "_t")) { CK(hipMemsetAsync(dbg, 0, 256, st)); dv.workspace = dbg; dv.workspace_bytes = 256; }
    if (v.fn(dv, st)) { printf("  %s rc!=0\n", v.name); continue; }
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) v.fn(dv, st);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("  %-16s %8.1f us  %7.1f TF/s\n", v.name, ms / reps * 1e3, 2.0 * M * N * K / (ms / reps) / 1e9);
    if (strstr(v.name, "_t")) {
      unsigned h[64];
      CK(hipMemcpy(h, dbg, 256, hipMemcpyDeviceToHost));
      for (int wv = 0; wv < 8; wv += 4) {
        double tot = 0; for (int i = 0; i < 5; ++i) tot += h[wv * 8 + i];
        printf("    [wave %d] %%: vmwait %.1f barrier1 %.1f load %.1f barrier2 %.1f multiply %.1f\n", wv, 100 * h[wv * 8] / tot, 100 * h[wv * 8 + 1] / tot,
               100 * h[wv * 8 + 2] / tot, 100 * h[wv * 8 + 3] / tot, 100 * h[wv * 8 + 4] / tot);
      }
    }
  }
  return 0;
}

static int run_tn_suite(hipStream_t st, int reps) {
  struct Shape { int M, N, K; };
  const Shape shapes[] = {{3072, 768, 55296}, {768, 3072, 55296}, {2304, 768, 55296}, {768, 768, 55296}, {2048, 512, 50176}, {512, 2048, 50176}, {1536, 512, 50176},
                          {512, 512, 50176}, {768, 3072, 13824}, {1024, 4096, 12544}, {1000, 520, 8192}};
  void* ws; const size_t wsb = (size_t)192 << 20;
  CK(hipMalloc(&ws, wsb));
  for (const Shape& s : shapes) {
    const int M = s.M, N = s.N, K = s.K;
    const int Mp = (M + 7) & ~7, Np = (N + 7) & ~7;
    u16 *A, *B; float *C, *cs;
    CK(hipMalloc(&A, (size_t)K * Mp * 2)); CK(hipMalloc(&B, (size_t)K * Np * 2)); CK(hipMalloc(&C, (size_t)M * Np * 4)); CK(hipMalloc(&cs, (size_t)Mp * 4));
    fill_kernel<<<2048, 256, 0, st>>>(A, (size_t)K * Mp, 0x1234u);
    fill_kernel<<<2048, 256, 0, st>>>(B, (size_t)K * Np, 0x9876u);
    const int NR = 32;
    std::vector<int> rows(NR);
    for (int i = 0; i < NR; ++i) rows[i] = (int)(((long)i * 7919 * 13) % M);
    rows[0] = 0; rows[1] = M - 1; rows[2] = std::min(M - 1, 255); rows[3] = std::min(M - 1, 256); rows[4] = std::min(M - 1, 127); rows[5] = std::min(M - 1, 128);
    int* drows; float* R;
    CK(hipMalloc(&drows, NR * 4)); CK(hipMalloc(&R, (size_t)NR * N * 4));
    CK(hipMemcpyAsync(drows, rows.data(), NR * 4, hipMemcpyHostToDevice, st));
    ref_rows_tn_kernel<<<dim3((N + 255) / 256, NR), 256, 0, st>>>(A, B, drows, R, Mp, Np, K);
    std::vector<float> hR((size_t)NR * N);
    CK(hipMemcpyAsync(hR.data(), R, hR.size() * 4, hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
    vmvm_gemm_desc d;
    memset(&d, 0, sizeof(d));
    d.A = A; d.B = B; d.C = C; d.M = M; d.N = N; d.K = K; d.lda = Mp; d.ldb = Np; d.ldc = Np; d.a_kmajor = 0; d.b_kmajor = 0;
    d.col_scale = 1.f; d.out_fp32 = 1; d.accumulate = 1; d.workspace = ws; d.workspace_bytes = (int64_t)wsb; d.colsum = cs;
    printf("TN M=%d N=%d K=%d (f32 accumulate, split-K slabs + reduce, fused column sums)\n", M, N, K);
    const char* names[2] = {"old128", "auto"};
    for (int v = 0; v < 2; ++v) {
      d.variant = v == 0 ? 6 : 0;
      CK(hipMemsetAsync(C, 0, (size_t)M * Np * 4, st)); CK(hipMemsetAsync(cs, 0, (size_t)Mp * 4, st));
      int rc = vmvm_gemm_bf16(&d, st);
      if (rc) { printf("  %-8s rc=%d\n", names[v], rc); continue; }
      CK(hipStreamSynchronize(st));
      std::vector<float> hC((size_t)NR * N), hcs(M);
      for (int r = 0; r < NR; ++r) CK(hipMemcpyAsync(hC.data() + (size_t)r * N, C + (size_t)rows[r] * Np, N * 4, hipMemcpyDeviceToHost, st));
      CK(hipMemcpyAsync(hcs.data(), cs, M * 4, hipMemcpyDeviceToHost, st));
      CK(hipStreamSynchronize(st));
      double me = 0;
      for (size_t i = 0; i < hC.size(); ++i) {
        const double err = fabs((double)hC[i] - hR[i]) / (fabs((double)hR[i]) * 1e-3 + 1e-3 * sqrt((double)K));
        if (!(err <= me)) me = err;
      }
      // column sums: |sum_k A[k][m]| -- checked against a host sum of the sampled columns
      std::vector<u16> col(K);
      double mcs = 0;
      for (int r = 0; r < 4; ++r) {
        CK(hipMemcpy2D(col.data(), 2, A + rows[r], (size_t)Mp * 2, 2, K, hipMemcpyDeviceToHost));
        double sref = 0;
        for (int k = 0; k < K; ++k) { uint32_t b = (uint32_t)col[k] << 16; float f; memcpy(&f, &b, 4); sref += f; }
        mcs = std::max(mcs, fabs(sref - hcs[rows[r]]) / (1e-3 * sqrt((double)K) + 1e-3 * fabs(sref)));
        if (getenv("PROBE_VERBOSE")) printf("    colsum m=%d ref %.4f got %.4f\n", rows[r], sref, hcs[rows[r]]);
      }
      std::vector<float> tms;
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      for (int rd = 0; rd < 3; ++rd) {
        vmvm_gemm_bf16(&d, st);
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) vmvm_gemm_bf16(&d, st);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        tms.push_back(ms / reps);
      }
      std::sort(tms.begin(), tms.end());
      printf("  %-8s med %8.1f us  %7.1f TF/s  relerr %.3f %s  colsum err %.3f %s\n", names[v], tms[1] * 1e3, 2.0 * M * N * K / tms[1] / 1e9, me, me <= 1.0 ? "ok" : "MISMATCH",
             mcs, mcs <= 1.0 ? "ok" : "MISMATCH");
      fflush(stdout);
    }
    CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C)); CK(hipFree(cs)); CK(hipFree(drows)); CK(hipFree(R));
  }
  return 0;
}

struct Variant { const char* name; int (*fn)(const vmvm_gemm_desc&, hipStream_t); };

static int run_old(const vmvm_gemm_desc& d, hipStream_t st) { vmvm_gemm_desc e = d; e.variant = 6; return vmvm_gemm_bf16(&e, st); }
static int run_auto(const vmvm_gemm_desc& d, hipStream_t st) { return vmvm_gemm_bf16(&d, st); }

template <int F>
static std::vector<Variant> variants() {
  std::vector<Variant> v;
  v.push_back({"old128", run_old});
  v.push_back({"auto", run_auto});
  v.push_back({"pp2_32x4", launch_pp_f<true, true, F, 2, 32, 4>});
  v.push_back({"pp2_64x2", launch_pp_f<true, true, F, 2, 64, 2>});
  v.push_back({"pp2_64x2_ne", launch_pp_f<true, true, F, 2, 64, 2, 1>});
  v.push_back({"pp2_q64", launch_pp_f<true, true, F, 2, 64, 2, 8>});            // a quarter of the CUs (grid 64): per-tile time against the full grid's
  v.push_back({"pp2_q64_ne", launch_pp_f<true, true, F, 2, 64, 2, 9>});
  return v;
}

static float host_gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678f)); }

int main(int argc, char** argv) {
  const char* set = argc > 1 ? argv[1] : "step";
  const int reps = argc > 2 ? atoi(argv[2]) : 10;
  const char* vfilter = argc > 3 ? argv[3] : "";          // comma-free substring filter on the variant name ("" = all)
  const int rounds = argc > 4 ? atoi(argv[4]) : 3;
  struct Shape { int M, N, K; };
  std::vector<Shape> shapes;
  if (!strcmp(set, "small")) shapes = {{512, 512, 256}, {1000, 520, 128}, {4096, 4096, 4096}};
  else if (!strcmp(set, "big")) shapes = {{8192, 8192, 8192}, {4096, 4096, 4096}};
  else if (!strcmp(set, "k8")) shapes = {{8192, 8192, 8192}};
  else if (!strcmp(set, "fc1")) shapes = {{69120, 3072, 768}};
  else if (!strcmp(set, "roof")) shapes = {{55296, 3072, 768}, {50176, 2048, 512}};
  else if (!strcmp(set, "swin1")) shapes = {{802816, 512, 128}, {802816, 128, 512}, {802816, 384, 128}, {200704, 1024, 256}};
  else if (!strcmp(set, "stag")) shapes = {{69120, 3072, 768}, {69120, 2304, 768}, {65536, 768, 3072}, {50176, 1536, 512}};
  else if (!strcmp(set, "step2")) shapes = {{55296, 3072, 768}, {55296, 768, 3072}, {55296, 2304, 768}, {55296, 768, 2304}, {55296, 768, 768}, {13824, 3072, 768},
                                            {13824, 768, 3072}, {13824, 2304, 768}, {13824, 768, 768}, {50176, 512, 512}, {12544, 1024, 4096}, {12544, 3072, 1024}};
  else shapes = {{69120, 3072, 768}, {69120, 768, 3072}, {69120, 2304, 768}, {69120, 768, 768}, {50176, 2048, 512}, {50176, 512, 2048}, {50176, 1536, 512},
                 {12544, 4096, 1024}, {8192, 8192, 8192}};
  hipStream_t st;
  CK(hipStreamCreate(&st));
  if (!strcmp(set, "tn")) return run_tn_suite(st, reps);
  if (!strcmp(set, "tnloop")) return run_tn_mainloop(st, reps);
  for (const Shape& s : shapes) {
    const int M = s.M, N = s.N, K = s.K;
    u16 *A, *B, *C, *C2;
    float* bias;
    CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&B, (size_t)N * K * 2));
    CK(hipMalloc(&C, (size_t)M * N * 2)); CK(hipMalloc(&C2, (size_t)M * N * 2)); CK(hipMalloc(&bias, (size_t)N * 4));
    fill_kernel<<<2048, 256, 0, st>>>(A, (size_t)M * K, 0x1234u);
    fill_kernel<<<2048, 256, 0, st>>>(B, (size_t)N * K, 0x9876u);
    fill_f32_kernel<<<(N + 255) / 256, 256, 0, st>>>(bias, N, 77u);
    // sampled reference rows (first / last rows of tiles, edges)
    const int NR = 64;
    std::vector<int> rows(NR);
    for (int i = 0; i < NR; ++i) rows[i] = (int)(((long)i * 7919 * 131 + (i % 3 == 0 ? 0 : 255 * i)) % M);
    rows[0] = 0; rows[1] = M - 1; rows[2] = std::min(M - 1, 255); rows[3] = std::min(M - 1, 256); rows[4] = std::min(M - 1, 127); rows[5] = std::min(M - 1, 128);
    int* drows; float* R;
    CK(hipMalloc(&drows, NR * 4)); CK(hipMalloc(&R, (size_t)NR * N * 4));
    CK(hipMemcpyAsync(drows, rows.data(), NR * 4, hipMemcpyHostToDevice, st));
    ref_rows_kernel<<<dim3((N + 255) / 256, NR), 256, 0, st>>>(A, B, drows, R, N, K, K, K);
    std::vector<float> hR((size_t)NR * N), hb(N);
    CK(hipMemcpyAsync(hR.data(), R, hR.size() * 4, hipMemcpyDeviceToHost, st));
    CK(hipMemcpyAsync(hb.data(), bias, N * 4, hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));

    u16* X;                                              // aux / residual operand [M][N]
    CK(hipMalloc(&X, (size_t)M * N * 2));
    fill_kernel<<<2048, 256, 0, st>>>(X, (size_t)M * N, 0x4242u);
    std::vector<u16> hX((size_t)NR * N);
    for (int r = 0; r < NR; ++r) CK(hipMemcpyAsync(hX.data() + (size_t)r * N, X + (size_t)rows[r] * N, N * 2, hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
    for (int epi = 0; epi < 4; ++epi) {
      vmvm_gemm_desc d;
      memset(&d, 0, sizeof(d));
      d.A = A; d.B = B; d.C = C; d.M = M; d.N = N; d.K = K; d.lda = K; d.ldb = K; d.ldc = N; d.a_kmajor = 1; d.b_kmajor = 1;
      d.col_scale = 1.f; d.splitk = 1;
      // PROBE_HOT_A / PROBE_HOT_B: row stride 8 elements -> the operand's rows overlap and the whole operand is ~1 MB, i.e. L2-resident for the
      // whole launch (results are garbage): what would the kernel do if that operand never missed L2?
      if (getenv("PROBE_HOT_A")) d.lda = 8;
      if (getenv("PROBE_HOT_B")) d.ldb = 8;
      const bool c8 = getenv("PROBE_CODE8") != nullptr;      // 8-bit GELU' codes: C2 / aux are byte tensors (the act-3 reference below does not apply then)
      if (epi == 1) { d.bias = bias; d.act = 1; d.C2 = C2; d.ldc2 = N; d.aux_code8 = c8; }
      if (epi == 2) { d.act = 3; d.aux = X; d.ldaux = N; d.aux_code8 = c8; }
      if (epi == 3) { d.bias = bias; d.resid = X; d.ldr = N; }
      std::vector<Variant> vs = epi == 0 ? variants<0>() : epi == 1 ? variants<(EF_BIAS | EF_ACT1 | EF_RS)>() : epi == 2 ? variants<(EF_ACT3 | EF_RS)>()
                                                                                                 : variants<(EF_BIAS | EF_RESID | EF_DROP | EF_RS | EF_MAP)>();
      if (vfilter[0]) {
        std::vector<Variant> keep;
        for (auto& v : vs) if (strstr(v.name, vfilter)) keep.push_back(v);
        vs = keep;
      }
      const int nv = (int)vs.size();
      std::vector<std::vector<float>> tms(nv);
      std::vector<double> maxerr(nv, 0.0);
      std::vector<int> rc(nv, 0);
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      // correctness first
      std::vector<u16> hC((size_t)NR * N);
      unsigned* dbg;
      CK(hipMalloc(&dbg, 8 * 8 * 4));
      for (int v = 0; v < nv; ++v) {
        CK(hipMemsetAsync(C, 0xff, (size_t)M * N * 2, st));
        const bool timers = strstr(vs[v].name, "_t") != nullptr;
        vmvm_gemm_desc dv = d;
        if (timers) { CK(hipMemsetAsync(dbg, 0, 256, st)); dv.workspace = dbg; dv.workspace_bytes = 256; }
        rc[v] = vs[v].fn(dv, st);
        if (rc[v]) continue;
        CK(hipStreamSynchronize(st));
        if (timers) {
          unsigned h[64];
          CK(hipMemcpy(h, dbg, 256, hipMemcpyDeviceToHost));
          for (int wv = 0; wv < 8; wv += 4) {
            double tot = 0;
            for (int i = 0; i < 5; ++i) tot += h[wv * 8 + i];
            printf("  [%s wave %d] cycles: vmwait %u  barrier1 %u  load %u  barrier2 %u  multiply+loop %u   (%%: %.1f %.1f %.1f %.1f %.1f)\n", vs[v].name, wv,
                   h[wv * 8], h[wv * 8 + 1], h[wv * 8 + 2], h[wv * 8 + 3], h[wv * 8 + 4], 100 * h[wv * 8] / tot, 100 * h[wv * 8 + 1] / tot,
                   100 * h[wv * 8 + 2] / tot, 100 * h[wv * 8 + 3] / tot, 100 * h[wv * 8 + 4] / tot);
          }
        }
        if (strstr(vs[v].name, "_ne")) continue;
        for (int r = 0; r < NR; ++r) CK(hipMemcpyAsync(hC.data() + (size_t)r * N, C + (size_t)rows[r] * N, N * 2, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
        double me = 0;
        for (int r = 0; r < NR; ++r)
          for (int n = 0; n < N; ++n) {
            float want = hR[(size_t)r * N + n];
            if (epi == 1) want = host_gelu(want + hb[n]);
            uint32_t xb = (uint32_t)hX[(size_t)r * N + n] << 16;
            float xv; memcpy(&xv, &xb, 4);
            if (epi == 2) want *= 0.5f * (1.0f + erff(xv * 0.70710678f)) + xv * 0.39894228f * expf(-0.5f * xv * xv);
            if (epi == 3) want = want + hb[n] + xv;
            uint32_t bits = (uint32_t)hC[(size_t)r * N + n] << 16;
            float got; memcpy(&got, &bits, 4);
            const double err = fabs((double)got - want) / (fabs((double)want) * 0.008 + 0.02 * sqrt((double)K) * 0.05 + 1e-3);   // bf16 rounding + accumulation order
            if (!(err <= me)) me = err;
          }
        maxerr[v] = me;
      }
      // timing: interleaved rounds
      for (int rd = 0; rd < rounds; ++rd)
        for (int v = 0; v < nv; ++v) {
          if (rc[v]) continue;
          vs[v].fn(d, st);
          CK(hipEventRecord(e0, st));
          for (int i = 0; i < reps; ++i) vs[v].fn(d, st);
          CK(hipEventRecord(e1, st));
          CK(hipEventSynchronize(e1));
          float ms;
          CK(hipEventElapsedTime(&ms, e0, e1));
          tms[v].push_back(ms / reps);
        }
      const char* enames[4] = {"plain", "bias+gelu+pre", "act3+aux", "bias+resid"};
      printf("M=%d N=%d K=%d epi=%s\n", M, N, K, enames[epi]);
#ifdef VMVM_PROBE_TIMELINE
      for (int v = 0; v < nv; ++v)
        if (!rc[v] && !strcmp(vs[v].name, "old128") && (epi == 1 || epi == 0)) {
          CK(hipStreamSynchronize(st));
          probe_timeline_reset();
          for (int i = 0; i < 20; ++i) vs[v].fn(d, st);       // warm: the timeline kept is the LAST launch's
          CK(hipStreamSynchronize(st));
          probe_timeline_dump(1);
        }
#endif
      for (int v = 0; v < nv; ++v) {
        if (rc[v]) { printf("  %-8s rc=%d\n", vs[v].name, rc[v]); continue; }
        std::sort(tms[v].begin(), tms[v].end());
        const double flop = 2.0 * M * N * K;
        printf("  %-8s med %8.1f us  %7.1f TF/s  (best %7.1f)  relerr %.3f %s\n", vs[v].name, tms[v][rounds / 2] * 1e3, flop / tms[v][rounds / 2] / 1e9,
               flop / tms[v][0] / 1e9, maxerr[v], maxerr[v] <= 1.0 ? "ok" : "MISMATCH");
      }
      fflush(stdout);
    }
    CK(hipFree(X));
    CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C)); CK(hipFree(C2)); CK(hipFree(bias)); CK(hipFree(drows)); CK(hipFree(R));
  }
  return 0;
}
