#!/usr/bin/env python3
"""Development harness: run every libvmvm kernel against a plain torch reference on the GPU and print error
summaries WITHOUT stopping at the first failure (one gpurun trip -> maximum information).
The pytest suite (tests/test_kernels_gpu.py) asserts the same properties."""
import math
import os
import sys
import traceback

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_empirical_mvm_amd import kernels as K  # noqa: E402
from pytorch_empirical_mvm_amd import swin_index as SI  # noqa: E402

dev = "cuda"
BF, F32 = torch.bfloat16, torch.float32
RESULTS = []


def rep(name, got, ref, tol=2e-2):
    got, ref = got.float(), ref.float()
    err = (got - ref).abs().max().item()
    scale = ref.abs().max().item() + 1e-12
    bad = not math.isfinite(err) or err / scale > tol
    RESULTS.append((name, err, scale, bad))
    print(f"{'FAIL' if bad else 'ok  '} {name:55s} maxerr={err:.4e} refmax={scale:.4e} rel={err / scale:.3e}", flush=True)
    return not bad


def run(fn):
    try:
        fn()
    except Exception:
        RESULTS.append((fn.__name__, float('nan'), 0, True))
        print(f"EXC  {fn.__name__}")
        traceback.print_exc()
    torch.cuda.synchronize()


def rnd(*shape, scale=1.0, dtype=BF):
    return (torch.randn(*shape, device=dev) * scale).to(dtype)


# ------------------------------------------------------------------ probe
def check_probe():
    m = K.probe_tr16()
    exp = torch.zeros(64, 4, dtype=torch.int32)
    for lane in range(64):
        g, i = lane >> 4, lane & 15
        for e in range(4):
            exp[lane, e] = (4 * g + e) * 16 + i
    ok = torch.equal(m, exp)
    print("probe tr16 mapping as assumed:", ok)
    if not ok:
        print(m[:20])
    RESULTS.append(("probe_tr16", 0.0, 1.0, not ok))


# ------------------------------------------------------------------ gemm
def check_gemm_layouts():
    for (M, N, K_) in [(392 * 3, 384, 128), (1000, 30528, 768), (256, 128, 96), (130, 72, 64), (4096, 512, 2048)]:
        A = rnd(M, K_)
        B = rnd(N, K_)
        ref = A.float() @ B.float().t()
        rep(f"gemm NT {M}x{N}x{K_}", K.gemm(A, B), ref)
        if M % 8 == 0 and N % 8 == 0:
            At, Bt = A.t().contiguous(), B.t().contiguous()
            for var in (0, 1):
                rep(f"gemm NN(b n-major) v{var} {M}x{N}x{K_}", K.gemm(A, Bt, b_kmajor=False, variant=var), ref)
                rep(f"gemm TN(a m-major,b n-major) v{var} {M}x{N}x{K_}", K.gemm(At, Bt, a_kmajor=False, b_kmajor=False, variant=var), ref)
                rep(f"gemm TT(a m-major) v{var} {M}x{N}x{K_}", K.gemm(At, B, a_kmajor=False, variant=var), ref)


def check_gemm_colsum():
    """bias gradient fused into the weight-gradient GEMM (colsum of the m-major A operand) + the unfused fallback paths"""
    for (rows, n_out, k_in, ws) in [(4096, 768, 768, True), (50176, 512, 2048, True), (5000, 200, 136, False), (13824, 2304, 768, True), (1000, 30528, 768, True)]:
        K._WORKSPACE.clear()
        if ws:
            K.set_workspace(torch.empty(256 << 20, device=dev, dtype=torch.uint8))
        dy, x = rnd(rows, n_out), rnd(rows, k_in)
        gw = torch.randn(n_out, k_in, device=dev)
        gb = torch.randn(n_out, device=dev)
        gw0, gb0 = gw.clone(), gb.clone()
        K.gemm(dy, x, a_kmajor=False, b_kmajor=False, M=n_out, N=k_in, K=rows, out=gw, accumulate=True, colsum=gb)
        rep(f"wgrad+colsum dW rows={rows} {n_out}x{k_in} ws={ws}", gw, gw0 + dy.float().t() @ x.float())
        rep(f"wgrad+colsum db rows={rows} {n_out}x{k_in} ws={ws}", gb, gb0 + dy.float().sum(0))
        # colsum_scale: the ONE DropPath scale of a compact Swin branch rides on the fused form (and on the separate-pass fallback)
        gw, gb = gw0.clone(), gb0.clone()
        K.gemm(dy, x, a_kmajor=False, b_kmajor=False, M=n_out, N=k_in, K=rows, out=gw, accumulate=True, colsum=gb, colsum_scale=1.25)
        rep(f"wgrad+colsum*1.25 dW rows={rows} {n_out}x{k_in} ws={ws}", gw, gw0 + dy.float().t() @ x.float())
        rep(f"wgrad+colsum*1.25 db rows={rows} {n_out}x{k_in} ws={ws}", gb, gb0 + 1.25 * dy.float().sum(0))
    K._WORKSPACE.clear()


def check_gemm_fp16_conv():
    """fp16 builds of the persistent GEMM + implicit 3x3 convolution (frozen dVAE tokenizer) against torch in fp32"""
    H16 = torch.float16
    def r16(*sh): return (torch.randn(*sh, device=dev) * 0.5).to(H16)
    for (M, N, K_) in [(4096, 256, 512), (1000, 136, 192), (6272, 8192, 2048)]:
        A, Bw, bias = r16(M, K_), r16(N, K_), torch.randn(N, device=dev)
        ref = A.float() @ Bw.float().t() + bias
        rep(f"gemm fp16 {M}x{N}x{K_} bias", K.gemm(A, Bw, bias=bias, fp16=True).float(), ref)
        rep(f"gemm fp16 {M}x{N}x{K_} bias+relu", K.gemm(A, Bw, bias=bias, act=2, fp16=True).float(), torch.relu(ref))
        res = r16(M, N)
        rep(f"gemm fp16 {M}x{N}x{K_} (xW+b)*g+res", K.gemm(A, Bw, bias=bias, col_scale=0.25, col_scale_n=N, resid=res, fp16=True).float(), ref * 0.25 + res.float())
        rep(f"gemm fp16 {M}x{N}x{K_} f32 out", K.gemm(A, Bw, bias=bias, out_dtype=torch.float32, fp16=True), ref, tol=2e-3)
    for (n, Hh, Ww, Ci, Co) in [(3, 14, 10, 64, 128), (2, 28, 28, 128, 256), (5, 56, 56, 64, 64), (1, 7, 9, 256, 72)]:
        x = r16(n, Hh, Ww, Ci)                                                        # NHWC
        w = (torch.randn(Co, Ci, 3, 3, device=dev) / (3 * Ci ** 0.5)).to(H16)
        bias = torch.randn(Co, device=dev) * 0.1
        w2 = w.permute(0, 2, 3, 1).reshape(Co, 9 * Ci).contiguous()                   # k = tap * C_in + c
        ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).float(), w.float(), bias, padding=1).permute(0, 2, 3, 1).reshape(-1, Co)
        y = K.gemm(x.view(-1, Ci), w2, bias=bias, fp16=True, conv=(9, Hh, Ww))
        rep(f"conv3x3 fp16 n={n} {Hh}x{Ww} {Ci}->{Co}", y.float(), ref)
        y = K.gemm(x.view(-1, Ci), w2, bias=bias, act=2, fp16=True, conv=(9, Hh, Ww))
        rep(f"conv3x3 fp16 n={n} {Hh}x{Ww} {Ci}->{Co} relu", y.float(), torch.relu(ref))


def check_dvae_passes():
    """round-2 passes of the dVAE tokenizer: A-operand ReLU, the 256x64 tile (64 output channels), fused arg-max, NHWC max-pool, stem
    im2col with the pixel pre-processing, against torch in fp32"""
    H16 = torch.float16
    def r16(*sh): return (torch.randn(*sh, device=dev) * 0.5).to(H16)
    for (n, Hh, Ww, Ci, Co) in [(3, 14, 10, 64, 64), (2, 28, 28, 256, 64), (1, 9, 7, 128, 48), (2, 12, 12, 64, 128)]:
        x = r16(n, Hh, Ww, Ci)
        w = (torch.randn(Co, Ci, 3, 3, device=dev) / (3 * Ci ** 0.5)).to(H16)
        bias = torch.randn(Co, device=dev) * 0.1
        w2 = w.permute(0, 2, 3, 1).reshape(Co, 9 * Ci).contiguous()
        for ar in (False, True):
            xin = torch.relu(x) if ar else x
            ref = torch.relu(torch.nn.functional.conv2d(xin.permute(0, 3, 1, 2).float(), w.float(), bias, padding=1)).permute(0, 2, 3, 1).reshape(-1, Co)
            y = K.gemm(x.view(-1, Ci), w2, bias=bias, act=2, fp16=True, conv=(9, Hh, Ww), a_relu=ar)
            rep(f"conv3x3 fp16 n={n} {Hh}x{Ww} {Ci}->{Co} relu a_relu={int(ar)}", y.float(), ref)
    for (M, N, K_) in [(1000, 512, 256), (3000, 8192, 512), (777, 128, 128)]:
        A, Bw, bias = r16(M, K_), r16(N, K_), torch.randn(N, device=dev)
        ref = torch.relu(A).float() @ Bw.float().t() + bias
        rep(f"gemm fp16 a_relu {M}x{N}x{K_} f32 out", K.gemm(A, Bw, bias=bias, out_dtype=torch.float32, fp16=True, a_relu=True), ref, tol=2e-3)
        groups = N // 64
        pairs = torch.empty((M, 2 * groups), device=dev, dtype=torch.float32)
        K.gemm(A, Bw, bias=bias, out=pairs, N=N, act=5, fp16=True, a_relu=True)
        tok = K.argmax_pairs(pairs, groups)
        lg = K.gemm(A, Bw, bias=bias, out_dtype=torch.float32, fp16=True, a_relu=True)
        rep(f"fused arg-max {M}x{N}x{K_} == argmax(logits)", tok.float(), torch.argmax(lg, 1).float(), tol=0.0)
        rep(f"fused arg-max {M}x{N}x{K_} pair maxima", pairs.view(M, groups, 2)[:, :, 0].max(1).values, lg.max(1).values, tol=1e-6)
    for (n, Hh, Ww, C_) in [(3, 8, 12, 64), (2, 56, 56, 256), (1, 2, 2, 8)]:
        x = r16(n * Hh * Ww, C_)
        ref = torch.nn.functional.max_pool2d(x.view(n, Hh, Ww, C_).permute(0, 3, 1, 2).float(), 2).permute(0, 2, 3, 1).reshape(-1, C_)
        rep(f"maxpool2x2 nhwc n={n} {Hh}x{Ww}x{C_}", K.maxpool2x2_nhwc(x, n, Hh, Ww).float(), ref, tol=0.0)
    for (n, Hh, Ww) in [(2, 16, 24), (1, 9, 5)]:
        img = torch.randn(n, 3, Hh, Ww, device=dev)
        w = torch.randn(32, 3, 7, 7, device=dev) / 12.0
        mean = torch.tensor([0.485, 0.456, 0.406], device=dev).view(1, 3, 1, 1)
        std = torch.tensor([0.229, 0.224, 0.225], device=dev).view(1, 3, 1, 1)
        xp = 0.8 * (img * std + mean) + 0.1
        ref = torch.nn.functional.conv2d(xp.half().float(), w.half().float(), None, padding=3).permute(0, 2, 3, 1).reshape(-1, 32)
        wk = torch.zeros((32, 7, 8, 3), device=dev)
        wk[:, :, :7, :] = w.permute(0, 2, 3, 1)
        wk = torch.nn.functional.pad(wk.reshape(32, 168), (0, 24)).half()
        cols = K.dvae_stem_im2col(img)
        rep(f"stem im2col n={n} {Hh}x{Ww} (via the 7x7 conv)", cols.float() @ wk.float().t(), ref, tol=1e-3)


def check_pool_grad():
    """gradient of the token pool (gather-sum through the static video fan-in and the text CSR) against index_add in fp32"""
    import numpy as np
    for (B, O, Lv, X, Hd, third) in [(4, 4, 20, 8, 64, False), (5, 3, 7, 4, 128, True), (2, 2, 3, 2, 8, False)]:
        Lq = Lv + X
        g1, g2 = rnd(B * Lq, Hd), rnd(B * O * Lq, Hd)
        g3 = rnd(B * Lq, Hd) if third else None
        tj = np.concatenate([[i] + list(np.random.permutation([j for j in range(B) if j != i])[:O - 1]) for i in range(B)]).astype(np.int64)
        order = np.argsort(tj, kind="stable")
        csr = torch.from_numpy(np.concatenate([[0], np.cumsum(np.bincount(tj, minlength=B)), order]).astype(np.int32)).to(dev)
        out = K.pool_grad(g1, g2, g3, B, O, Lv, X, csr[:B + 1], csr[B + 1:])
        ref = torch.zeros(B * Lv + B * X, Hd, device=dev)
        ar_v, ar_t = np.arange(Lv), np.arange(X)
        idx1 = torch.from_numpy(np.concatenate([np.concatenate([i * Lv + ar_v, B * Lv + i * X + ar_t]) for i in range(B)])).to(dev)
        idx2 = torch.from_numpy(np.concatenate([np.concatenate([(p_ // O) * Lv + ar_v, B * Lv + int(tj[p_]) * X + ar_t]) for p_ in range(B * O)])).to(dev)
        ref.index_add_(0, idx1, g1.float())
        ref.index_add_(0, idx2, g2.float())
        if third:
            ref.index_add_(0, idx1, g3.float())
        rep(f"pool_grad B={B} O={O} Lv={Lv} X={X} Hd={Hd} third={int(third)}", out.float(), ref, tol=8e-3)


def check_gemm_big(variant=4, tag="big"):
    """256^2-tile kernel (variant 4) / 3-stage 256x128 kernel (variant 5) on every layout + epilogue paths + split-K."""
    for (M, N, K_) in [(392 * 3, 768, 128), (1000, 1024, 768), (4096, 512, 2048), (777 * 8, 256, 64)]:
        A, B = rnd(M, K_), rnd(N, K_)
        ref = A.float() @ B.float().t()
        At, Bt = A.t().contiguous(), B.t().contiguous()
        rep(f"{tag} NT {M}x{N}x{K_}", K.gemm(A, B, variant=variant), ref)
        rep(f"{tag} NN {M}x{N}x{K_}", K.gemm(A, Bt, b_kmajor=False, variant=variant), ref)
        rep(f"{tag} TN {M}x{N}x{K_}", K.gemm(At, Bt, a_kmajor=False, b_kmajor=False, variant=variant), ref)
        rep(f"{tag} TT {M}x{N}x{K_}", K.gemm(At, B, a_kmajor=False, variant=variant), ref)
    M, N, K_ = 2 * 392 * 2, 512, 256
    A, B = rnd(M, K_), rnd(N, K_, scale=0.1)
    bias = torch.randn(N, device=dev)
    base = A.float() @ B.float().t()
    pre = torch.empty(M, N, device=dev, dtype=BF)
    rep(tag + " gelu", K.gemm(A, B, bias=bias, act=1, out_preact=pre, variant=variant), torch.nn.functional.gelu(base + bias))
    rep(tag + " gelu preact", pre, base + bias)
    perm = torch.randperm(392, device=dev).int()
    perm[::50] = -1
    resid = rnd(4 * 392, N)
    out = torch.zeros(4 * 392, N, device=dev, dtype=BF)
    rs = torch.rand(4, device=dev) + 0.5
    K.gemm(A, B, bias=bias, resid=resid, row_map=perm, map_len=392, map_stride=392, out=out, row_scale=rs, rows_per_scale=392, scale_bias_only=True, variant=variant)
    ref = torch.zeros(4 * 392, N, device=dev)
    pl = perm.long()
    for b in range(4):
        ok = pl >= 0
        ref[b * 392 + pl[ok]] = (base + bias * rs[b])[b * 392:(b + 1) * 392][ok] + resid.float()[b * 392 + pl[ok]]
    rep(tag + " row_map + resid + scaled bias", out, ref)
    ws = torch.empty(64 << 20, device=dev, dtype=torch.uint8)
    for sk, w in ((0, ws), (5, ws), (3, None)):
        acc = torch.randn(N, K_, device=dev)
        acc0 = acc.clone()
        dy = rnd(M, N)
        K.gemm(dy, A, a_kmajor=False, b_kmajor=False, M=N, N=K_, K=M, out=acc, accumulate=True, splitk=sk, workspace=w, variant=variant)
        rep(f"{tag} wgrad splitk={sk} ws={w is not None}", acc, dy.float().t() @ A.float() + acc0, tol=2e-3)


def check_gemm_p3():
    check_gemm_big(5, "p3")


def check_code8(variant, M, N, K_, tag):
    """8-bit GELU' code (vmvm_gemm_desc.aux_code8): forward output unchanged, code within one step of the f32 rounding of GELU'(pre),
    backward = multiply by the decode; pre-activations spread over +-6 so both saturating ends and the negative lobe of GELU' are hit"""
    A, B = rnd(M, K_), rnd(N, K_, scale=0.1)
    base = A.float() @ B.float().t()
    bias = torch.randn(N, device=dev)
    Aw, Bw = rnd(M, K_, scale=1.0), rnd(N, K_, scale=2.0 / math.sqrt(K_))
    basew = Aw.float() @ Bw.float().t()
    code = torch.full((M, N), 7, device=dev, dtype=torch.uint8)
    rs8 = torch.rand((M + 195) // 196, device=dev) + 0.5
    rows8 = torch.arange(M, device=dev) // 196
    out8 = K.gemm(Aw, Bw, bias=bias, act=1, out_preact=code, code8=True, row_scale=rs8, rows_per_scale=196, variant=variant)
    rep(f"{tag} {M}x{N}x{K_} gelu (code8 build)", out8, torch.nn.functional.gelu(basew + bias) * rs8[rows8, None])
    pf = (basew + bias).requires_grad_(True)
    torch.nn.functional.gelu(pf).sum().backward()
    want = torch.round((pf.grad + 0.13) * (255.0 / 1.26)).clamp(0, 255)
    dcode = (code.float() - want).abs()
    print(f"     code8: max |code - round(g)| = {dcode.max().item():.0f}, off-by-one fraction {(dcode > 0).float().mean().item():.4f}, "
          f"code range {int(code.min())}..{int(code.max())}")
    # (round 4: GELU / GELU' share one exponential -- a logistic form within 6.7e-4 of the erf derivative, an eighth of a code step, so a
    #  code sits one step off the rounding of the exact derivative for ~6 % of the elements; never more than one step)
    RESULTS.append((f"{tag} code8 codes within 1 of the f32 rounding", dcode.max().item(), 1, dcode.max().item() > 1 or (dcode > 0).float().mean().item() > 0.10))
    dec = code.float() * (1.26 / 255.0) - 0.13
    rep(f"{tag} code8 decode vs GELU'", dec, pf.grad, tol=0.0033 / 1.13)
    rep(f"{tag} act3 code8", K.gemm(A, B, act=3, aux=code, code8=True, row_scale=rs8, rows_per_scale=196, variant=variant), base * dec * rs8[rows8, None])
    rep(f"{tag} act3 code8 vs exact GELU'", K.gemm(A, B, act=3, aux=code, code8=True, variant=variant), base * pf.grad, tol=1e-2)


def check_gemm_epilogues():
    M, N, K_ = 784, 256, 128
    A, B = rnd(M, K_), rnd(N, K_, scale=0.1)
    bias = torch.randn(N, device=dev)
    base = A.float() @ B.float().t()
    rep("gemm bias", K.gemm(A, B, bias=bias), base + bias)
    pre = torch.empty(M, N, device=dev, dtype=BF)
    out = K.gemm(A, B, bias=bias, act=1, out_preact=pre)
    rep("gemm gelu", out, torch.nn.functional.gelu(base + bias))
    rep("gemm gelu preact", pre, base + bias)
    rep("gemm relu", K.gemm(A, B, bias=bias, act=2), torch.relu(base + bias))
    # the GELU form itself against torch.nn.functional.gelu (erf) in fp32: pre-activations on a grid over [-9, 9] through an identity
    # weight (exact in bf16 up to the grid's rounding), so the only differences are the logistic approximation (<= 3.4e-4 absolute,
    # csrc/common.h gelu_and_code2) and the bf16 rounding of the output (2^-9 relative); the 8-bit code against the erf derivative
    xs = torch.linspace(-9, 9, 256 * 128, device=dev).to(BF).view(256, 128)
    eye = torch.eye(128, device=dev, dtype=BF)
    want = torch.nn.functional.gelu(xs.float())
    for tagv, kw in (("", {}), (" (code8 build)", dict(out_preact=torch.empty(256, 128, device=dev, dtype=torch.uint8), code8=True))):
        got = K.gemm(xs, eye, act=1, **kw).float()
        err = ((got - want).abs() - want.abs() * 2.0 ** -8).clamp_min(0).max().item()
        print(f"     gelu form{tagv}: max |error| beyond the bf16 rounding of the output = {err:.2e} (bound 3.5e-4)")
        RESULTS.append((f"gemm gelu vs torch erf-GELU fp32{tagv}", err, 3.5e-4, err > 3.5e-4))
        if kw:
            xf = xs.float().requires_grad_(True)
            torch.nn.functional.gelu(xf).sum().backward()
            dec = kw["out_preact"].float() * (1.26 / 255.0) - 0.13
            derr = (dec - xf.grad).abs().max().item()
            print(f"     gelu' code: max |decode - erf derivative| = {derr:.2e} (half a code step 2.5e-3 + form 6.7e-4)")
            RESULTS.append(("gemm gelu' code vs torch erf-GELU' fp32", derr, 3.3e-3, derr > 3.3e-3))
    u = rnd(M, N)
    uf = u.float().requires_grad_(True)
    torch.nn.functional.gelu(uf).sum().backward()
    rep("gemm act3 gelu'", K.gemm(A, B, act=3, aux=u), base * uf.grad)
    rep("gemm act4 relu'", K.gemm(A, B, act=4, aux=u), base * (u.float() > 0))
    check_code8(0, 784, 256, 128, 'pers')
    check_code8(7, 1000, 560, 256, 'pp')
    check_code8(7, 4096, 3072, 768, 'pp')
    rs = torch.rand(4, device=dev) + 0.5
    rows = torch.arange(M, device=dev) // 196
    rep("gemm row_scale", K.gemm(A, B, bias=bias, row_scale=rs, rows_per_scale=196), (base + bias) * rs[rows, None])
    rep("gemm scale_bias_only", K.gemm(A, B, bias=bias, row_scale=rs, rows_per_scale=196, scale_bias_only=True), base + bias * rs[rows, None])
    r = rnd(M, N)
    rep("gemm resid", K.gemm(A, B, resid=r), base + r.float())
    rep("gemm fp32 out", K.gemm(A, B, out_dtype=F32), base, tol=1e-3)
    acc = torch.randn(M, N, device=dev)
    acc0 = acc.clone()
    K.gemm(A, B, out=acc, accumulate=True)
    rep("gemm fp32 accumulate", acc, base + acc0, tol=1e-3)
    ws = torch.empty(64 << 20, device=dev, dtype=torch.uint8)
    for sk, w in ((1, None), (3, None), (0, None), (3, ws), (0, ws), (7, ws)):
        acc = torch.randn(N, K_, device=dev)
        acc0 = acc.clone()
        K.gemm(A.t().contiguous().t().contiguous(), A, a_kmajor=False, b_kmajor=False, M=K_, N=K_, K=M, out=acc[:K_], accumulate=True, splitk=sk, workspace=w)
        rep(f"gemm wgrad-shape accumulate splitk={sk} ws={w is not None}", acc[:K_], A.float().t() @ A.float() + acc0[:K_], tol=2e-3)
    rep("gemm col_scale", K.gemm(A, B, bias=bias, col_scale=0.25, col_scale_n=64),
        torch.cat([(base + bias)[:, :64] * 0.25, (base + bias)[:, 64:]], 1))
    # row map: 2 clips x 392 slots -> token order, with pads
    perm = torch.randperm(392, device=dev).int()
    perm[::50] = -1
    resid = rnd(2 * 392, N)
    out = torch.zeros(2 * 392, N, device=dev, dtype=BF)
    K.gemm(A, B, resid=resid, row_map=perm, map_len=392, map_stride=392, out=out)
    ref = torch.zeros(2 * 392, N, device=dev)
    for b in range(2):
        for s in range(392):
            d = int(perm[s])
            if d >= 0:
                ref[b * 392 + d] = base[b * 392 + s] + resid[b * 392 + d].float()
    rep("gemm row_map + resid", out, ref)
    # dropout: keep fraction and scaling
    o = K.gemm(A, B, dropout_p=0.1, seed=1234, offset=7).float()
    kept = (o != 0)
    frac = 1.0 - kept.float().mean().item()
    print(f"     dropout drop fraction {frac:.4f} (want 0.1)")
    rep("gemm dropout kept values", torch.where(kept, o, torch.zeros_like(o)), torch.where(kept, base / 0.9, torch.zeros_like(base)))
    o2 = K.gemm(A, B, dropout_p=0.1, seed=1234, offset=7).float()
    RESULTS.append(("gemm dropout deterministic", 0, 1, not torch.equal(o, o2) or abs(frac - 0.1) > 0.01))


def check_gemm_round_split():
    """vmvm_gemm_bf16's whole-round split (rows that fill whole rounds of the 256-workgroup grid on the ping-pong kernel, the rest on
    the 128x128 kernel): the output -- dropout mask, bias, residual included -- equals the one-launch result of the ping-pong kernel
    (variant 7) at a shape that takes the split: 69120 x 768 x 2048 = 810 tiles = 3.16 rounds."""
    M, N, K_ = 69120, 768, 2048
    A, B = rnd(M, K_), rnd(N, K_, scale=0.05)
    bias = torch.randn(N, device=dev)
    r = rnd(M, N)
    kw = dict(bias=bias, resid=r, dropout_p=0.1, seed=99, offset=4242)
    one = K.gemm(A, B, variant=7, **kw)
    two = K.gemm(A, B, **kw)
    d1, d2 = K.gemm(A, B, variant=7, dropout_p=0.1, seed=99, offset=4242), K.gemm(A, B, dropout_p=0.1, seed=99, offset=4242)
    same_mask = torch.equal(d1 == 0, d2 == 0) and abs(float((d2[65536:] == 0).float().mean()) - 0.1) < 0.01      # (no bias / residual: zero <=> dropped)
    RESULTS.append(("gemm round split: dropout mask identical to the one-launch result", 0, 1, not same_mask))
    rep("gemm round split vs one launch (bias + dropout + resid)", two, one.float())
    idx = torch.tensor([0, 1, 65535, 65536, 65537, 69119], device=dev)
    ref = (A[idx].float() @ B.float().t() + bias)
    keep = ((one[idx].float() - r[idx].float()).abs() > 1e-6).float()
    rep("gemm round split rows around the split vs fp32", two[idx], ref * keep / 0.9 + r[idx].float())
    rep("gemm round split plain", K.gemm(A, B), K.gemm(A, B, variant=7).float())
    # round 5: per-clip row scales move with the rows (vmvm_gemm_desc.scale_row0): the Swin stage-3 fc2 shape, 30 clips of 1 568 rows
    # (the split point, 32 768 rows, falls inside clip 20), DropPath 'producer' form (the scale multiplies the bias only) and the plain form
    Bc, Lc = 30, 1568
    M2, N2, K2 = Bc * Lc, 512, 2048
    A2, B2 = rnd(M2, K2), rnd(N2, K2, scale=0.05)
    bias2, r2 = torch.randn(N2, device=dev), rnd(M2, N2)
    rs2 = torch.rand(Bc, device=dev) + 0.5
    for sbo in (True, False):
        kw2 = dict(bias=bias2, resid=r2, row_scale=rs2, rows_per_scale=Lc, scale_bias_only=sbo)
        got = K.gemm(A2, B2, **kw2)
        acc = A2.float() @ B2.float().t()
        rsr = rs2.repeat_interleave(Lc)[:, None]
        ref2 = (acc + bias2 * rsr if sbo else (acc + bias2) * rsr) + r2.float()
        rep(f"gemm round split with per-clip row scales (scale_bias_only={sbo}) vs fp32", got, ref2)
        rep(f"gemm round split with per-clip row scales (scale_bias_only={sbo}) vs the 128x128 kernel", got, K.gemm(A2, B2, variant=6, **kw2).float(), tol=1e-2)          # (two kernels: one bf16 ulp of the largest outputs)
    # round 6: the ABSOLUTE row-map form (a DropPath-compacted Swin fc2: the rows of the kept clips scatter into the full tensor, padding
    # clips carry -1; residual read through the same map).  28 kept clips of 32 (+ 2 padding clips) -> 30 x 1 568 rows.  (Measured on
    # this shape and not kept: the ping-pong kernel + whole-round split with the map moving with the rows -- 141.6 us against the 128x128
    # kernel's 125.5.)
    Bt = 32
    kept = torch.randperm(Bt, device=dev)[:28].sort().values
    lst = torch.cat([kept, torch.full((2,), -1, device=dev, dtype=kept.dtype)]).to(torch.int32)
    amap = torch.where(lst[:, None] >= 0, lst[:, None] * Lc + torch.arange(Lc, device=dev, dtype=torch.int32)[None, :], torch.full((1, 1), -1, device=dev, dtype=torch.int32)).reshape(-1).contiguous().to(torch.int32)
    x1 = rnd(Bt * Lc, N2)
    rs3 = torch.cat([torch.full((28,), 1.0 / 0.9, device=dev), torch.zeros(2, device=dev)])
    kw3 = dict(bias=bias2, resid=x1, row_scale=rs3, rows_per_scale=Lc, scale_bias_only=True, row_map=amap, map_len=Bc * Lc, map_stride=0, out_rows=Bt * Lc)
    got3 = K.gemm(A2, B2, **kw3)
    old3 = K.gemm(A2, B2, variant=6, **kw3)
    rows = amap[amap >= 0].long()
    src_rows = torch.nonzero(amap >= 0).squeeze(1)
    ref3 = (A2[src_rows].float() @ B2.float().t() + bias2 / 0.9) + x1[rows].float()
    rep("gemm absolute row map (compacted fc2: bias x scale + residual through the map) vs fp32", got3[rows], ref3)
    rep("gemm absolute row map: auto dispatch vs the 128x128 kernel", got3[rows], old3[rows].float(), tol=1e-2)


# ------------------------------------------------------------------ layernorm
def check_ln():
    for C_, with_ws in ((96, False), (128, True), (256, True), (256, False), (512, False), (512, True), (768, True), (768, False), (1024, True), (2048, True), (3072, False), (3072, True)):
        # with_ws: dgamma/dbeta partials through the scratch buffer + column-reduce kernel ; without: global atomics
        K._WORKSPACE.clear()
        if with_ws:
            K.set_workspace(torch.empty(64 << 20, device=dev, dtype=torch.uint8))
        M = 777 if C_ != 768 else 5000
        x = rnd(M, C_)
        g = torch.randn(C_, device=dev) * 0.1 + 1
        b = torch.randn(C_, device=dev) * 0.1
        y, mean, rstd = K.layernorm_fwd(x, g, b, 1e-5)
        xf = x.float().requires_grad_(True)
        gf, bf = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
        ref = torch.nn.functional.layer_norm(xf, (C_,), gf, bf, 1e-5)
        rep(f"ln fwd C={C_}", y, ref)
        dy = rnd(M, C_)
        ref.backward(dy.float())
        dg, db = torch.zeros(C_, device=dev), torch.zeros(C_, device=dev)
        add = rnd(M, C_)
        dx, dx2 = K.layernorm_bwd(dy, x, g, mean, rstd, dg, db, dX_add=add, want_dX2=True, dropout_p=0.1, seed=5, offset=3)
        rep(f"ln bwd dx(+add) C={C_}", dx, xf.grad + add.float())
        rep(f"ln bwd dgamma C={C_} ws={with_ws}", dg, gf.grad)
        rep(f"ln bwd dbeta C={C_} ws={with_ws}", db, bf.grad)
        if C_ == 768 and with_ws:
            # dX2 must equal the GEMM-epilogue dropout mask pattern (same seed/offset, N=C)
            A = torch.eye(128, device=dev, dtype=BF)
            ones = K.gemm(torch.ones(M, 128, device=dev, dtype=BF), torch.ones(C_, 128, device=dev, dtype=BF), dropout_p=0.1, seed=5, offset=3)
            mask = (ones.float() != 0)
            rep("ln bwd dX2 mask consistent with gemm dropout", dx2, torch.where(mask, dx.float() / 0.9, torch.zeros_like(dx.float())))
    K._WORKSPACE.clear()


def check_ln_gather():
    # window map (with padding + shift) and merge map
    B, D, H, W, C_ = 2, 12, 24, 20, 64
    ws, ss = SI.get_window_size((D, H, W), (8, 7, 7), (4, 3, 3))
    m, (Dp, Hp, Wp) = SI.window_map(D, H, W, ws, ss)
    Lp, Lq = m.size, D * H * W
    src = torch.from_numpy(m).to(dev)
    x = rnd(B * Lq, C_)
    g = torch.randn(C_, device=dev) * 0.1 + 1
    b = torch.randn(C_, device=dev) * 0.1
    y, mean, rstd = K.layernorm_fwd(x, g, b, 1e-5, M=B * Lp, C_=C_, nseg=1, src=src, rows_out_per_batch=Lp, rows_in_per_batch=Lq, pad_mode=0)
    xf = x.float().requires_grad_(True)
    gf, bf = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ln = torch.nn.functional.layer_norm(xf, (C_,), gf, bf, 1e-5).view(B, Lq, C_)
    srcl = src.long()
    ref = torch.where((srcl >= 0)[None, :, None], ln[:, srcl.clamp(min=0)], torch.zeros((), device=dev)).reshape(B * Lp, C_)
    rep("ln window-gather fwd", y, ref)
    dy = rnd(B * Lp, C_)
    ref.backward(dy.float())
    dg, db = torch.zeros(C_, device=dev), torch.zeros(C_, device=dev)
    add = rnd(B * Lq, C_)
    dx, _ = K.layernorm_bwd(dy, x, g, mean, rstd, dg, db, rows_in=B * Lq, nseg=1, src=src, rows_out_per_batch=Lp, rows_in_per_batch=Lq,
                            pad_mode=0, dX_add=add)
    rep("ln window-gather bwd dx(+add)", dx, xf.grad + add.float())
    rep("ln window-gather bwd dgamma", dg, gf.grad)
    rep("ln window-gather bwd dbeta", db, bf.grad)
    # the same backward walked in SOURCE order through the inverse map (vmvm_ln_bwd_desc.inv): identical rows, identical results; and with
    # an absolute map of a subset of the clips (the engine's compact form): rows of the clips left out stay untouched
    inv = K.invert_map(src, Lq)
    RESULTS.append(("invert_map is the inverse of the window map", float(((src[inv.long()] != torch.arange(Lq, device=dev)) & (inv >= 0)).sum() + (inv < 0).sum()), 0.0,
                    bool((inv < 0).any() or (src[inv.long()] != torch.arange(Lq, device=dev)).any())))
    dg2, db2 = torch.zeros(C_, device=dev), torch.zeros(C_, device=dev)
    dx2, _ = K.layernorm_bwd(dy, x, g, mean, rstd, dg2, db2, rows_in=B * Lq, nseg=1, src=src, rows_out_per_batch=Lp, rows_in_per_batch=Lq,
                             pad_mode=0, dX_add=add, inv=inv)
    rep("ln window-gather bwd, source-major: dx identical", dx2, dx, tol=0)
    rep("ln window-gather bwd, source-major: dgamma", dg2, gf.grad)
    rep("ln window-gather bwd, source-major: dbeta", db2, bf.grad)
    src_abs = torch.cat([src, torch.full_like(src, -1)])            # clip 0 kept, one padding clip; absolute rows (clip 1 of the input is left out)
    inv_abs = K.invert_map(src_abs, B * Lq)
    dxa = torch.full((B * Lq, C_), 7.0, device=dev, dtype=BF)
    dg3, db3 = torch.zeros(C_, device=dev), torch.zeros(C_, device=dev)
    K.layernorm_bwd(torch.cat([dy[:Lp], torch.zeros_like(dy[:Lp])]), x, g, torch.cat([mean[:Lp], mean[:Lp]]), torch.cat([rstd[:Lp], rstd[:Lp]]), dg3, db3, dX=dxa, rows_in=B * Lq, nseg=1,
                    src=src_abs, rows_out_per_batch=2 * Lp, rows_in_per_batch=B * Lq, pad_mode=0, dX_add=add, inv=inv_abs)
    rep("ln window-gather bwd, source-major, absolute map: kept clip", dxa[:Lq], dx[:Lq], tol=0)
    rep("ln window-gather bwd, source-major, absolute map: other clip untouched", dxa[Lq:], torch.full_like(dxa[Lq:], 7.0), tol=0)
    # round 5: scattered dX of the identity walk (dx_map = a per-batch permutation) and the residual gradient by OUTPUT row (add_by_out):
    # bit-identical to "ordinary backward, then permute the rows" / "gather the residual gradient first" -- un-padded window maps, the
    # packed (C = 64, 128, 256) and the row-per-wave (C = 512) kernels
    for (Cq, dims_q) in ((64, (8, 14, 14)), (128, (8, 14, 7)), (256, (8, 7, 14)), (512, (8, 14, 14))):
        Dq, Hq, Wq = dims_q
        wsq, ssq = SI.get_window_size(dims_q, (8, 7, 7), (0, 3, 3))
        mq, _ = SI.window_map(Dq, Hq, Wq, wsq, ssq)
        Lw = Dq * Hq * Wq
        assert mq.size == Lw and (mq >= 0).all()
        srcq = torch.from_numpy(mq).to(dev)
        invq = K.invert_map(srcq, Lw)
        xq = rnd(B * Lw, Cq)
        gq = torch.randn(Cq, device=dev) * 0.1 + 1
        bq = torch.randn(Cq, device=dev) * 0.1
        # identity LayerNorm (norm2 of a block): dX through dx_map
        yq, mq_, rq_ = K.layernorm_fwd(xq, gq, bq, 1e-5)
        dyq, addq = rnd(B * Lw, Cq), rnd(B * Lw, Cq)
        d0, d1 = torch.zeros(Cq, device=dev), torch.zeros(Cq, device=dev)
        dxn, _ = K.layernorm_bwd(dyq, xq, gq, mq_, rq_, d0, d1, dX_add=addq)
        e0, e1 = torch.zeros(Cq, device=dev), torch.zeros(Cq, device=dev)
        dxw, _ = K.layernorm_bwd(dyq, xq, gq, mq_, rq_, e0, e1, dX_add=addq, dx_map=invq)
        rep(f"ln bwd dx_map C={Cq}: rows = gather of the ordinary dX", dxw, K.gather_rows(dxn, srcq, B * Lw, Lw, Lw), tol=0)
        rep(f"ln bwd dx_map C={Cq}: dgamma", e0, d0, tol=1e-5)            # (partial sums meet through atomics: order-dependent last bits)
        # gathered LayerNorm (norm1): residual gradient by output row
        yg, mg_, rg_ = K.layernorm_fwd(xq, gq, bq, 1e-5, M=B * Lw, C_=Cq, nseg=1, src=srcq, rows_out_per_batch=Lw, rows_in_per_batch=Lw, pad_mode=0)
        f0, f1 = torch.zeros(Cq, device=dev), torch.zeros(Cq, device=dev)
        dxa_, _ = K.layernorm_bwd(dyq, xq, gq, mg_, rg_, f0, f1, rows_in=B * Lw, nseg=1, src=srcq, rows_out_per_batch=Lw, rows_in_per_batch=Lw, pad_mode=0, dX_add=dxn)
        h0, h1 = torch.zeros(Cq, device=dev), torch.zeros(Cq, device=dev)
        dxb_, _ = K.layernorm_bwd(dyq, xq, gq, mg_, rg_, h0, h1, rows_in=B * Lw, nseg=1, src=srcq, rows_out_per_batch=Lw, rows_in_per_batch=Lw, pad_mode=0, dX_add=dxw, add_by_out=True)
        rep(f"ln bwd add_by_out C={Cq}: identical to the source-row form", dxb_, dxa_, tol=0)
    # gather_rows (window_partition of a gradient)
    gr = K.gather_rows(x, src, B * Lp, Lp, Lq)
    refg = torch.where((srcl >= 0)[None, :, None], x.float().view(B, Lq, C_)[:, srcl.clamp(min=0)], torch.zeros((), device=dev)).reshape(B * Lp, C_)
    rep("gather_rows window", gr, refg, tol=0)
    # merge
    D, H, W = 3, 7, 5
    mm, (D2, H2, W2) = SI.merge_map(D, H, W)
    srcm = torch.from_numpy(mm).to(dev)
    Lq, Lo = D * H * W, D2 * H2 * W2
    x = rnd(B * Lq, C_)
    g4 = torch.randn(4 * C_, device=dev) * 0.1 + 1
    b4 = torch.randn(4 * C_, device=dev) * 0.1
    y, mean, rstd = K.layernorm_fwd(x, g4, b4, 1e-5, M=B * Lo, C_=4 * C_, nseg=4, src=srcm, rows_out_per_batch=Lo, rows_in_per_batch=Lq, pad_mode=1)
    xf = x.float().requires_grad_(True)
    gf, bf = g4.clone().requires_grad_(True), b4.clone().requires_grad_(True)
    sl = srcm.long().view(Lo, 4)
    cat = torch.where((sl >= 0)[None, :, :, None], xf.view(B, Lq, C_)[:, sl.clamp(min=0)], torch.zeros((), device=dev)).reshape(B * Lo, 4 * C_)
    ref = torch.nn.functional.layer_norm(cat, (4 * C_,), gf, bf, 1e-5)
    rep("ln merge-gather fwd", y, ref)
    dy = rnd(B * Lo, 4 * C_)
    ref.backward(dy.float())
    dg, db = torch.zeros(4 * C_, device=dev), torch.zeros(4 * C_, device=dev)
    dx, _ = K.layernorm_bwd(dy, x, g4, mean, rstd, dg, db, rows_in=B * Lq, nseg=4, src=srcm, rows_out_per_batch=Lo, rows_in_per_batch=Lq, pad_mode=1)
    rep("ln merge-gather bwd dx", dx, xf.grad)
    rep("ln merge-gather bwd dgamma", dg, gf.grad)


# ------------------------------------------------------------------ attention
def attn_ref(q, k, v, bias, drop_mask=None):
    s = q @ k.transpose(-1, -2) + bias
    p = s.softmax(-1)
    if drop_mask is not None:
        p = p * drop_mask
    return p @ v


def check_attn_window():
    for (dims, B, heads) in [((8, 14, 14), 2, 4), ((4, 7, 7), 3, 3), ((12, 24, 20), 1, 2)]:
        D, H, W = dims
        win = (8, 7, 7)
        for shifted, layout in ((False, 0), (True, 0), (False, 1), (True, 1)):
            ws, ss = SI.get_window_size(dims, win, (4, 3, 3) if shifted else (0, 0, 0))
            if layout and not SI.win3_ok(ws, ss):
                continue
            m, (Dp, Hp, Wp) = SI.window_map(D, H, W, ws, ss)
            N = ws[0] * ws[1] * ws[2]
            nW = m.size // N
            reg = SI.region_ids(Dp, Hp, Wp, ws, ss)
            rc, rc0 = SI.rc_codes(N, win)
            if layout:                                    # the win_layout = 1 token order: slot tables permuted, the reference below is order-agnostic
                pm = SI.win3_perm()
                rc = np.ascontiguousarray(rc[pm])
                reg = None if reg is None else np.ascontiguousarray(reg[:, pm])
            C_ = heads * 32
            nseq = B * nW
            qkv = rnd(nseq * N, 3 * C_, scale=1.0)
            table = (torch.randn((2 * 8 - 1) * 13 * 13, heads, device=dev) * 0.5)
            rc_t = torch.from_numpy(rc).to(dev)
            reg_t = torch.from_numpy(reg).to(dev) if reg is not None else None
            sscale = (torch.rand(B, device=dev) + 0.5)
            out, lse = K.attention_fwd(qkv, nseq, N, heads, 32, 0, 32 ** -0.5, q_off=0, k_off=C_, v_off=2 * C_, bias_table=table, rc=rc_t,
                                       rc0=rc0, region=reg_t, n_win=nW, seq_scale=sscale, seqs_per_scale=nW, win_layout=layout)
            qf = qkv.float().requires_grad_(True)
            tf = table.clone().requires_grad_(True)
            x = qf.view(nseq, N, 3, heads, 32).permute(2, 0, 3, 1, 4)
            idx = (rc_t[:, None] - rc_t[None, :] + rc0).long()
            bias = tf[idx.reshape(-1)].view(N, N, heads).permute(2, 0, 1)[None]
            if reg is not None:
                mk = torch.where(reg_t[:, :, None] != reg_t[:, None, :], -100.0, 0.0)     # (nW,N,N)
                bias = bias + mk.repeat(B, 1, 1)[:, None]
            o = attn_ref(x[0], x[1], x[2], bias)                                              # (nseq, heads, N, 32)
            sc = sscale.repeat_interleave(nW)[:, None, None, None]
            ref = (o * sc).transpose(1, 2).reshape(nseq * N, C_)
            tag = f"win attn {dims} shifted={shifted} layout={layout}"
            rep(tag + " fwd", out, ref)
            dout = rnd(nseq * N, C_)
            ref.backward(dout.float())
            dtab = torch.zeros_like(table)
            dqkv = K.attention_bwd(dout, qkv, out, lse, nseq, N, heads, 32, 0, 32 ** -0.5, q_off=0, k_off=C_, v_off=2 * C_, bias_table=table,
                                   rc=rc_t, rc0=rc0, region=reg_t, n_win=nW, seq_scale=sscale, seqs_per_scale=nW, dbias_table=dtab, win_layout=layout)
            gq = qf.grad.clone()
            gq[:, :C_] *= 32 ** -0.5                      # kernel returns d(q_linear) = scale * d(q_scaled)
            rep(tag + " bwd dq", dqkv[:, :C_], gq[:, :C_])
            rep(tag + " bwd dk", dqkv[:, C_:2 * C_], gq[:, C_:2 * C_])
            rep(tag + " bwd dv", dqkv[:, 2 * C_:], gq[:, 2 * C_:])
            rep(tag + " bwd dtable", dtab, tf.grad)


def check_attn_window_spike():
    """win_layout = 1 kernels with scores that OVERFLOW against a first-key-block softmax reference (attention_win4.hip keeps a fixed
    reference per query row and repeats a sequence with the exact row maximum when l / O come out non-finite): a few (query, key) pairs
    get q.k ~ 8 |q|^2 ~ 250 with the key in a LATE block, one of them in the first block (no overflow there), forward and backward."""
    dims, B, heads, win = (8, 14, 14), 2, 2, (8, 7, 7)
    D, H, W = dims
    for shifted in (False, True):
        ws, ss = SI.get_window_size(dims, win, (4, 3, 3) if shifted else (0, 0, 0))
        m, (Dp, Hp, Wp) = SI.window_map(D, H, W, ws, ss)
        N = ws[0] * ws[1] * ws[2]
        nW = m.size // N
        reg = SI.region_ids(Dp, Hp, Wp, ws, ss)
        rc, rc0 = SI.rc_codes(N, win)
        pm = SI.win3_perm()
        rc = np.ascontiguousarray(rc[pm])
        reg = None if reg is None else np.ascontiguousarray(reg[:, pm])
        C_ = heads * 32
        nseq = B * nW
        qkv = rnd(nseq * N, 3 * C_, scale=1.0)
        q3 = qkv.view(nseq, N, 3, heads, 32)
        # (sequence, head, query slot, key slot): same mask region needed for the pair to be live -> use neighbours inside one tile pair
        # (round 6: + spikes whose key lies in the LAST live block of a masked walk -- sequence 2 = the h-split window: class-A queries walk
        # classes A + B, key slot 13 * 16 + 5 sits in B's last tile; sequence 3 = the corner window: class-C queries walk class C only, key in
        # tile 19; sequence 1 = the w-split window: class-B query (tile 9) against class D's tile 24 = the last block of the {B, D} walk)
        for (sq, hh, qi, kj) in [(0, 0, 5, 390), (1, 1, 200, 201), (nseq - 1, 0, 391, 388), (2, 1, 17, 3), (3, 0, 300, 310),
                                 (2, 0, 40, 13 * 16 + 5), (3, 1, 14 * 16 + 2, 19 * 16 + 9), (1, 0, 9 * 16 + 1, 24 * 16 + 3)]:
            q3[sq, kj, 1, hh] = (q3[sq, qi, 0, hh].float() * (4.5 if shifted else 8.0)).to(BF)    # (shifted: cross terms with MASKED queries must stay << 100 -- the kernels skip masked pairs, the reference adds -100)
        table = (torch.randn((2 * 8 - 1) * 13 * 13, heads, device=dev) * 0.5)
        rc_t = torch.from_numpy(rc).to(dev)
        reg_t = torch.from_numpy(reg).to(dev) if reg is not None else None
        # THE BAND JUST BELOW OVERFLOW (round 6): a row whose maximum sits 87.3 .. 88.7 above its first key block's maximum gives
        # l in (2^126, 2^128): finite -- no retry by a "non-finite" test -- but 1 / l is a DENORMAL, which v_rcp_f32 flushes to zero:
        # the row came out as zeros with a correct lse (found by a random query against a spiked key; attention_win4.hip `finite`).
        # Directed: class-A queries (their first live block is slots 0..31 in every window type), one late key each (class A's last
        # tile, so the pair is live under every shift mask) placed so that q.k + bias = (first-block max) + 86.5 .. 89.5 in 0.25 steps.
        idx_all = (rc_t[:, None] - rc_t[None, :] + rc0).long()
        for j in range(13):
            sq, hh, qi, kj = (j * 3) % nseq, j % heads, 33 + 7 * j, 112 + j                 # query tiles 2..7, key tile 7 (class A)
            qv = q3[sq, qi, 0, hh].float()
            s0 = q3[sq, :32, 1, hh].float() @ qv + table[idx_all[qi, :32], hh]
            target = float(s0.max()) + 86.5 + 0.25 * j - float(table[idx_all[qi, kj], hh])
            q3[sq, kj, 1, hh] = (qv * (target / float(qv @ qv))).to(BF)
        out, lse = K.attention_fwd(qkv, nseq, N, heads, 32, 0, 32 ** -0.5, q_off=0, k_off=C_, v_off=2 * C_, bias_table=table, rc=rc_t,
                                   rc0=rc0, region=reg_t, n_win=nW, win_layout=1)
        qf = qkv.float().requires_grad_(True)
        tf = table.clone().requires_grad_(True)
        x = qf.view(nseq, N, 3, heads, 32).permute(2, 0, 3, 1, 4)
        idx = (rc_t[:, None] - rc_t[None, :] + rc0).long()
        bias = tf[idx.reshape(-1)].view(N, N, heads).permute(2, 0, 1)[None]
        if reg is not None:
            mk = torch.where(reg_t[:, :, None] != reg_t[:, None, :], -100.0, 0.0)
            bias = bias + mk.repeat(B, 1, 1)[:, None]
        s_ = x[0] @ x[1].transpose(-1, -2) + bias
        o = s_.softmax(-1) @ x[2]
        ref = o.transpose(1, 2).reshape(nseq * N, C_)
        ref_lse = torch.logsumexp(s_, -1)                                                 # (nseq, heads, N)
        tag = f"win attn spike shifted={shifted}"
        gap = (s_.detach().amax(-1) - s_.detach()[..., :32].amax(-1))                     # row maximum above the first 32 slots' maximum
        print(f"     (max score {s_.max().item():.1f}, rows with max > 100: {(s_.amax(-1) > 100).sum().item()}, rows with the maximum 87.3-88.7 above "
              f"slots 0..31: {int(((gap > 87.3) & (gap < 88.7)).sum())})")
        assert int(((gap > 87.3) & (gap < 88.7)).sum()) >= 3, "the near-overflow band must be populated"
        rep(tag + " fwd", out, ref)
        rep(tag + " lse", lse.view(nseq, heads, N), ref_lse, tol=1e-3)
        dout = rnd(nseq * N, C_)
        ref.backward(dout.float())
        dtab = torch.zeros_like(table)
        dqkv = K.attention_bwd(dout, qkv, out, lse, nseq, N, heads, 32, 0, 32 ** -0.5, q_off=0, k_off=C_, v_off=2 * C_, bias_table=table,
                               rc=rc_t, rc0=rc0, region=reg_t, n_win=nW, dbias_table=dtab, win_layout=1)
        gq = qf.grad.clone()
        gq[:, :C_] *= 32 ** -0.5
        rep(tag + " bwd dq", dqkv[:, :C_], gq[:, :C_])
        rep(tag + " bwd dk", dqkv[:, C_:2 * C_], gq[:, C_:2 * C_])
        rep(tag + " bwd dv", dqkv[:, 2 * C_:], gq[:, 2 * C_:])
        rep(tag + " bwd dtable", dtab, tf.grad)


def _win_problem(dims, B, heads, shifted, win=(8, 7, 7)):
    """tables of one win_layout = 1 window-attention problem (slot order of swin_index.win3_perm)"""
    D, H, W = dims
    ws, ss = SI.get_window_size(dims, win, (4, 3, 3) if shifted else (0, 0, 0))
    m, (Dp, Hp, Wp) = SI.window_map(D, H, W, ws, ss)
    N = ws[0] * ws[1] * ws[2]
    nW = m.size // N
    reg = SI.region_ids(Dp, Hp, Wp, ws, ss)
    rc, rc0 = SI.rc_codes(N, win)
    pm = SI.win3_perm()
    rc_t = torch.from_numpy(np.ascontiguousarray(rc[pm])).to(dev)
    reg_t = None if reg is None else torch.from_numpy(np.ascontiguousarray(reg[:, pm])).to(dev)
    return N, nW, rc_t, rc0, reg_t


def _win_ref(qkv, table, nseq, N, heads, rc_t, rc0, reg_t, B):
    """fp32 torch reference of WindowAttention3D (video_swin.py:147-172) with the reference's ADDITIVE -100 shift mask (:292-307)"""
    C_ = heads * 32
    qf = qkv.float().requires_grad_(True)
    tf = table.clone().requires_grad_(True)
    x = qf.view(nseq, N, 3, heads, 32).permute(2, 0, 3, 1, 4)
    idx = (rc_t[:, None] - rc_t[None, :] + rc0).long()
    bias = tf[idx.reshape(-1)].view(N, N, heads).permute(2, 0, 1)[None]
    if reg_t is not None:
        mk = torch.where(reg_t[:, :, None] != reg_t[:, None, :], -100.0, 0.0)
        bias = bias + mk.repeat(B, 1, 1)[:, None]
    s_ = x[0] @ x[1].transpose(-1, -2) + bias
    o = s_.softmax(-1) @ x[2]
    return qf, tf, s_, o.transpose(1, 2).reshape(nseq * N, C_)


def check_attn_window_mask_boundary():
    """The shift mask at its boundary.  The reference ADDS -100 to the logits of pairs in different regions
    (video_swin.py:304-306); the win_layout = 1 kernels SKIP those pairs (probability exactly 0).  The two agree while
    exp(s_masked - 100 - max_live) underflows the f32 / bf16 resolution of the row, i.e. while no masked RAW logit exceeds the row's live
    maximum by more than ~83 (documented: include/vmvm.h vmvm_attn_fwd_desc.region).  Here masked raw logits exceed the live
    row maximum by ~30 (far outside anything a trained network produces for a cross-region pair, still 50 below the leak): forward,
    lse and every gradient must equal the additive-mask reference."""
    dims, B, heads = (8, 14, 14), 2, 2
    N, nW, rc_t, rc0, reg_t = _win_problem(dims, B, heads, True)
    C_ = heads * 32
    nseq = B * nW
    qkv = rnd(nseq * N, 3 * C_, scale=0.5)                 # live logits ~ N(0, 8 * 0.0625): row maxima ~ 3-5
    q3 = qkv.view(nseq, N, 3, heads, 32)
    reg_h = reg_t.cpu().numpy()
    cases = []
    for (sq, hh, qi) in [(1, 0, 5), (2, 1, 100), (3, 0, 391), (nseq - 1, 1, 230)]:
        wtype = sq % nW
        other = np.flatnonzero(reg_h[wtype] != reg_h[wtype, qi])          # key slots in ANOTHER region of this window: masked for query qi
        assert other.size > 0
        kj = int(other[len(other) // 2])
        qv = q3[sq, qi, 0, hh].float()
        q3[sq, kj, 1, hh] = (qv * (40.0 / float(qv @ qv))).to(BF)         # raw logit q.k = 40 (+ bias): ~30 above the live maximum, 60 below the leak
        cases.append((sq, hh, qi, kj))
    table = (torch.randn((2 * 8 - 1) * 13 * 13, heads, device=dev) * 0.5)
    out, lse = K.attention_fwd(qkv, nseq, N, heads, 32, 0, 32 ** -0.5, q_off=0, k_off=C_, v_off=2 * C_, bias_table=table, rc=rc_t,
                               rc0=rc0, region=reg_t, n_win=nW, win_layout=1)
    qf, tf, s_, ref = _win_ref(qkv, table, nseq, N, heads, rc_t, rc0, reg_t, B)
    raw = s_.detach() + torch.where(reg_t[:, :, None] != reg_t[:, None, :], 100.0, 0.0).repeat(B, 1, 1)[:, None]      # logits before the mask
    live_max = s_.detach().amax(-1)
    exc = min(float(raw[sq, hh, qi, kj] - live_max[sq, hh, qi]) for (sq, hh, qi, kj) in cases)
    print(f"     (masked raw logits exceed their row's live maximum by >= {exc:.1f})")
    assert exc > 15.0, exc
    tag = "win attn mask boundary (masked logit > live max + 15)"
    rep(tag + " fwd", out, ref)
    rep(tag + " lse", lse.view(nseq, heads, N), torch.logsumexp(s_, -1), tol=1e-3)
    dout = rnd(nseq * N, C_)
    ref.backward(dout.float())
    dtab = torch.zeros_like(table)
    dqkv = K.attention_bwd(dout, qkv, out, lse, nseq, N, heads, 32, 0, 32 ** -0.5, q_off=0, k_off=C_, v_off=2 * C_, bias_table=table,
                           rc=rc_t, rc0=rc0, region=reg_t, n_win=nW, dbias_table=dtab, win_layout=1)
    gq = qf.grad.clone()
    gq[:, :C_] *= 32 ** -0.5
    rep(tag + " bwd dq", dqkv[:, :C_], gq[:, :C_])
    rep(tag + " bwd dk", dqkv[:, C_:2 * C_], gq[:, C_:2 * C_])
    rep(tag + " bwd dv", dqkv[:, 2 * C_:], gq[:, 2 * C_:])
    rep(tag + " bwd dtable", dtab, tf.grad)


def check_attn_window_nonfinite():
    """The win4 forward's overflow retry must TERMINATE on non-finite input: a NaN / inf query row makes l / O non-finite
    on the first walk AND on the retry with the exact maxima -- the loop is `given -> break`, so the wave leaves after the second walk.
    The poisoned rows come out non-finite, every other sequence (and every other head of the poisoned sequences) equals the reference."""
    dims, B, heads = (8, 14, 14), 2, 2
    for shifted in (False, True):
        N, nW, rc_t, rc0, reg_t = _win_problem(dims, B, heads, shifted)
        C_ = heads * 32
        nseq = B * nW
        qkv = rnd(nseq * N, 3 * C_, scale=1.0)
        q3 = qkv.view(nseq, N, 3, heads, 32)
        q3[1, 7, 0, 0, 3] = float("nan")                      # one query element of (sequence 1, head 0, slot 7)
        q3[nseq - 1, 391, 0, 1, 0] = float("inf")             # the odd tile's wave, head 1, last sequence
        table = (torch.randn((2 * 8 - 1) * 13 * 13, heads, device=dev) * 0.5)
        out, lse = K.attention_fwd(qkv, nseq, N, heads, 32, 0, 32 ** -0.5, q_off=0, k_off=C_, v_off=2 * C_, bias_table=table, rc=rc_t,
                                   rc0=rc0, region=reg_t, n_win=nW, win_layout=1)
        torch.cuda.synchronize()                              # (returns: the retry loop ended)
        _, _, _, ref = _win_ref(qkv, table, nseq, N, heads, rc_t, rc0, reg_t, B)
        o4, r4 = out.float().view(nseq, N, heads, 32), ref.detach().view(nseq, N, heads, 32)
        bad = torch.zeros(nseq, N, heads, dtype=torch.bool, device=dev)
        bad[1, 7, 0] = True
        bad[nseq - 1, 391, 1] = True
        tag = f"win attn non-finite input shifted={shifted}"
        ok_rows = ~bad
        rep(tag + " clean rows", o4[ok_rows], r4[ok_rows])
        nonfin = (~torch.isfinite(o4[bad])).any(-1).all()
        RESULTS.append((tag + " poisoned rows non-finite", 0.0 if bool(nonfin) else 1.0, 0.0, not bool(nonfin)))
        print(f"  {'ok ' if bool(nonfin) else 'BAD'} {tag} poisoned rows non-finite")


def check_attn_bert():
    for (nseq, Lq, heads) in [(3, 432, 12), (24, 432, 12), (2, 232, 4), (2, 100, 2)]:      # (24 x 12 = 288 items: more than one per CU -- the persistent one-pass backward walks several)
        Hd = heads * 64
        qkv = rnd(nseq * Lq, 3 * Hd, scale=1.0)
        km = torch.ones(nseq, Lq, dtype=torch.uint8, device=dev)
        for s in range(nseq):
            km[s, Lq - 5 * (s + 1):] = 0
        out, lse = K.attention_fwd(qkv, nseq, Lq, heads, 64, 1, 0.125, q_off=0, k_off=Hd, v_off=2 * Hd, keymask=km)
        qf = qkv.float().requires_grad_(True)
        x = qf.view(nseq, Lq, 3, heads, 64).permute(2, 0, 3, 1, 4)
        bias = torch.where(km.bool(), 0.0, float("-inf"))[:, None, None, :]
        o = attn_ref(x[0] * 0.125, x[1], x[2], bias)
        ref = o.transpose(1, 2).reshape(nseq * Lq, Hd)
        tag = f"bert attn nseq={nseq} L={Lq} h={heads}"
        rep(tag + " fwd", out, ref)
        dout = rnd(nseq * Lq, Hd)
        ref.backward(dout.float())
        dqkv = K.attention_bwd(dout, qkv, out, lse, nseq, Lq, heads, 64, 1, 0.125, q_off=0, k_off=Hd, v_off=2 * Hd, keymask=km)
        rep(tag + " bwd dq", dqkv[:, :Hd], qf.grad[:, :Hd])
        rep(tag + " bwd dk", dqkv[:, Hd:2 * Hd], qf.grad[:, Hd:2 * Hd])
        rep(tag + " bwd dv", dqkv[:, 2 * Hd:], qf.grad[:, 2 * Hd:])
    # dropout consistency: recover the mask from V = I trick, then compare fwd/bwd with the same mask
    nseq, Lq, heads = 1, 64, 1
    Hd = 64
    qkv = torch.zeros(nseq * Lq, 3 * Hd, device=dev, dtype=BF)
    qkv[:, 2 * Hd:] = torch.eye(64, device=dev, dtype=BF)           # V = I, q=k=0 -> P uniform = 1/64
    out, lse = K.attention_fwd(qkv, nseq, Lq, heads, 64, 1, 0.125, q_off=0, k_off=Hd, v_off=2 * Hd, dropout_p=0.1, seed=11, offset=5)
    pm = out.float() * 64.0                                          # = mask/(1-p_eff)
    frac = (pm == 0).float().mean().item()
    print(f"     attn dropout drop fraction {frac:.4f} (p = 6554/65536 = 0.10001)")
    mask = (pm != 0).float() * (65536.0 / (65536.0 - 6554.0))
    qkv = rnd(nseq * Lq, 3 * Hd)
    out, lse = K.attention_fwd(qkv, nseq, Lq, heads, 64, 1, 0.125, q_off=0, k_off=Hd, v_off=2 * Hd, dropout_p=0.1, seed=11, offset=5)
    qf = qkv.float().requires_grad_(True)
    x = qf.view(nseq, Lq, 3, heads, 64).permute(2, 0, 3, 1, 4)
    o = attn_ref(x[0] * 0.125, x[1], x[2], 0.0, mask[None, None])
    ref = o.transpose(1, 2).reshape(nseq * Lq, Hd)
    rep("bert attn dropout fwd (mask recovered)", out, ref)
    dout = rnd(nseq * Lq, Hd)
    ref.backward(dout.float())
    dqkv = K.attention_bwd(dout, qkv, out, lse, nseq, Lq, heads, 64, 1, 0.125, q_off=0, k_off=Hd, v_off=2 * Hd, dropout_p=0.1, seed=11, offset=5)
    rep("bert attn dropout bwd dq", dqkv[:, :Hd], qf.grad[:, :Hd])
    rep("bert attn dropout bwd dk", dqkv[:, Hd:2 * Hd], qf.grad[:, Hd:2 * Hd])
    rep("bert attn dropout bwd dv", dqkv[:, 2 * Hd:], qf.grad[:, 2 * Hd:])

    # stored dropout decisions (vmvm_attn_fwd_desc.drop_mask, L = 432 exact-tile kernels): the record decodes to the mask the forward applied
    # (torch reference with that mask), and forward / backward are BIT-identical with and without it
    nseq, Lq, heads, Hd = 3, 432, 12, 768
    qkv = rnd(nseq * Lq, 3 * Hd)
    km = torch.ones(nseq, Lq, dtype=torch.uint8, device=dev)
    km[1, 400:] = 0
    kw = dict(q_off=0, k_off=Hd, v_off=2 * Hd, keymask=km, dropout_p=0.1, seed=23, offset=77)
    dm = K.attention_drop_mask(nseq, Lq, heads, 64, 1, 0.1, dev)
    assert dm is not None and dm.numel() == nseq * heads * 27 * 27 * 8
    dm.fill_(-1)
    out0, lse0 = K.attention_fwd(qkv, nseq, Lq, heads, 64, 1, 0.125, **kw)
    out1, lse1 = K.attention_fwd(qkv, nseq, Lq, heads, 64, 1, 0.125, drop_mask=dm, **kw)
    RESULTS.append(("bert attn drop_mask fwd bit-identical", float((out0 != out1).sum() + (lse0 != lse1).sum()), 0.0, bool((out0 != out1).any() or (lse0 != lse1).any())))
    bits = ((dm.view(nseq * heads, 27, 27, 4, 2, 1).to(torch.int64) >> torch.arange(32, device=dev)) & 1)          # [sh, qt, t, j, half, 32]
    bits = bits.reshape(nseq * heads, 27, 27, 4, 4, 16)                                                              # [sh, qt, t, j, g, r]
    dropped = bits.permute(0, 1, 5, 2, 4, 3).reshape(nseq, heads, Lq, Lq).float()                                    # [.., q = 16 qt + r, key = 16 t + 4 g + j]
    frac = dropped.mean().item()
    RESULTS.append(("bert attn drop_mask drop fraction", frac, 0.1, abs(frac - 0.1) > 0.002))
    mask = (1.0 - dropped) * (65536.0 / (65536.0 - 6554.0))
    qf = qkv.float().requires_grad_(True)
    x = qf.view(nseq, Lq, 3, heads, 64).permute(2, 0, 3, 1, 4)
    kb = torch.where(km.bool(), 0.0, float("-inf"))[:, None, None, :]
    sc = (x[0] * 0.125) @ x[1].transpose(-1, -2) + kb
    o = (torch.softmax(sc, -1) * mask) @ x[2]
    ref = o.transpose(1, 2).reshape(nseq * Lq, Hd)
    rep("bert attn drop_mask fwd vs torch with the decoded mask", out1, ref)
    dout = rnd(nseq * Lq, Hd)
    ref.backward(dout.float())
    g0 = K.attention_bwd(dout, qkv, out1, lse1, nseq, Lq, heads, 64, 1, 0.125, **kw)
    g1 = K.attention_bwd(dout, qkv, out1, lse1, nseq, Lq, heads, 64, 1, 0.125, drop_mask=dm, **kw)
    RESULTS.append(("bert attn drop_mask bwd bit-identical", float((g0 != g1).sum()), 0.0, bool((g0 != g1).any())))
    rep("bert attn drop_mask bwd dq vs torch", g1[:, :Hd], qf.grad[:, :Hd])
    rep("bert attn drop_mask bwd dk vs torch", g1[:, Hd:2 * Hd], qf.grad[:, Hd:2 * Hd])
    rep("bert attn drop_mask bwd dv vs torch", g1[:, 2 * Hd:], qf.grad[:, 2 * Hd:])
    # masked keys anywhere in the sequence (not only the padded text tail), 26 sequences x 12 heads = 312 (sequence, head) units,
    # with and without dropout + record
    nseq2 = 26
    qkv2 = rnd(nseq2 * Lq, 3 * Hd)
    km2 = torch.ones(nseq2, Lq, dtype=torch.uint8, device=dev)
    km2[0, 37] = 0; km2[1, 200:216] = 0; km2[2, 0:3] = 0; km2[3, 120:330] = 0; km2[4, 431] = 0
    for s_ in range(5, nseq2):
        km2[s_, torch.randperm(Lq, device=dev)[: 3 * s_]] = 0
    for pdrop in (0.0, 0.1):
        kw2 = dict(q_off=0, k_off=Hd, v_off=2 * Hd, keymask=km2, dropout_p=pdrop, seed=5, offset=3)
        dm2 = K.attention_drop_mask(nseq2, Lq, heads, 64, 1, 0.1, dev) if pdrop else None
        o2, l2 = K.attention_fwd(qkv2, nseq2, Lq, heads, 64, 1, 0.125, drop_mask=dm2, **kw2)
        if pdrop:
            b2 = ((dm2.view(nseq2 * heads, 27, 27, 4, 2, 1).to(torch.int64) >> torch.arange(32, device=dev)) & 1).reshape(nseq2 * heads, 27, 27, 4, 4, 16)
            mask2 = (1.0 - b2.permute(0, 1, 5, 2, 4, 3).reshape(nseq2, heads, Lq, Lq).float()) * (65536.0 / (65536.0 - 6554.0))
        else:
            mask2 = 1.0
        qf2 = qkv2.float().requires_grad_(True)
        x2 = qf2.view(nseq2, Lq, 3, heads, 64).permute(2, 0, 3, 1, 4)
        sc2 = (x2[0] * 0.125) @ x2[1].transpose(-1, -2) + torch.where(km2.bool(), 0.0, float("-inf"))[:, None, None, :]
        ref2 = ((torch.softmax(sc2, -1) * mask2) @ x2[2]).transpose(1, 2).reshape(nseq2 * Lq, Hd)
        do2 = rnd(nseq2 * Lq, Hd)
        ref2.backward(do2.float())
        g2 = K.attention_bwd(do2, qkv2, o2, l2, nseq2, Lq, heads, 64, 1, 0.125, drop_mask=dm2, **kw2)
        tag2 = f"bert attn scattered key mask, 312 units, p={pdrop}"
        rep(tag2 + " fwd", o2, ref2)
        rep(tag2 + " bwd dq", g2[:, :Hd], qf2.grad[:, :Hd])
        rep(tag2 + " bwd dk", g2[:, Hd:2 * Hd], qf2.grad[:, Hd:2 * Hd])
        rep(tag2 + " bwd dv", g2[:, 2 * Hd:], qf2.grad[:, 2 * Hd:])
        del qf2, x2, sc2, ref2, mask2
    # a problem without a stored-decision build refuses the buffer
    try:
        K.attention_fwd(qkv[: 2 * 100], 2, 100, heads, 64, 1, 0.125, drop_mask=dm, q_off=0, k_off=Hd, v_off=2 * Hd, dropout_p=0.1, seed=1, offset=0)
        RESULTS.append(("bert attn drop_mask refused where unsupported", 1.0, 0.0, True))
    except RuntimeError:
        RESULTS.append(("bert attn drop_mask refused where unsupported", 0.0, 0.0, False))


def check_attn_query_row():
    """vmvm_attn_query_row_fwd / bwd (one query position per sequence: the VTM pass' last fusion layer) against torch fp32 with the
    kernel's own dropout mask (probs_drop / probs): ragged key masks, head_dim 64 and 32, L = 432 and an odd 77"""
    for (nseq, Lq, heads, hd, p) in [(5, 432, 12, 64, 0.1), (3, 77, 4, 32, 0.0), (2, 512, 2, 64, 0.25), (2, 2352, 12, 64, 0.1)]:
        Hd = heads * hd
        q = rnd(nseq, Hd)
        kv = rnd(nseq * Lq, 2 * Hd + 64)[:, 32:32 + 2 * Hd + 8]          # a row pitch that is not the width (ld_kv != 2H), offset columns
        k_off, v_off = 8, 8 + Hd
        km = torch.ones(nseq, Lq, dtype=torch.uint8, device=dev)
        for s_ in range(nseq):
            km[s_, Lq - 3 * s_ - 1:] = 0
        scale = hd ** -0.5
        out, pr, prd = K.attn_query_row_fwd(q, kv, nseq, Lq, heads, hd, scale, k_off=k_off, v_off=v_off, keymask=km, dropout_p=p, seed=7, offset=123)
        qf = q.float().view(nseq, heads, hd).requires_grad_(True)
        kvf = kv.float().clone().requires_grad_(True)
        kf = kvf[:, k_off:k_off + Hd].reshape(nseq, Lq, heads, hd).permute(0, 2, 1, 3)
        vf = kvf[:, v_off:v_off + Hd].reshape(nseq, Lq, heads, hd).permute(0, 2, 1, 3)
        sc = torch.einsum("shd,shjd->shj", qf, kf) * scale
        sc = sc.masked_fill(~km.bool()[:, None, :], float("-inf"))
        pref = torch.softmax(sc, -1)
        mult = torch.where(pr > 0, prd / pr.clamp_min(1e-30), torch.zeros_like(pr))      # the kernel's dropout multiplier (0 or 1 / keep)
        tag = f"attn query-row nseq={nseq} L={Lq} h={heads} hd={hd} p={p}"
        rep(tag + " probs", pr, pref.detach(), tol=2e-3)
        if p > 0:
            frac = float((mult[pr > 1e-12] == 0).float().mean())
            RESULTS.append((tag + " drop fraction", frac, p, abs(frac - p) > 0.03))
            RESULTS.append((tag + " kept scale", 0, 1, bool(((mult > 0) & ((mult - 1 / (1 - p)).abs() > 1e-3)).any())))
        o_ref = torch.einsum("shj,shjd->shd", pref * mult, vf).reshape(nseq, Hd)
        rep(tag + " out", out, o_ref.detach())
        dout = rnd(nseq, Hd)
        o_ref.backward(dout.float())
        dq, dkv = K.attn_query_row_bwd(dout, q, kv, pr, prd, nseq, Lq, heads, hd, scale, k_off=k_off, v_off=v_off)
        rep(tag + " dq", dq, qf.grad.reshape(nseq, Hd))
        rep(tag + " dk", dkv[:, k_off:k_off + Hd], kvf.grad[:, k_off:k_off + Hd])
        rep(tag + " dv", dkv[:, v_off:v_off + Hd], kvf.grad[:, v_off:v_off + Hd])


def check_attn_stream():
    """Streaming attention kernels (L > 448, SURVEY C5 shapes): Swin-L-384 windows (8,12,12) = 1152 tokens with the 15x23x23 bias
    table and shift mask; fusion sequences of 16 x 384^2 clips (2352 tokens) and a ragged 600; and -- the dropout stream being a
    function of the absolute (query, key) position only -- the streaming kernels forced at the step's own shapes (432 / 392
    tokens) against the resident kernels with the same seed."""
    win = (8, 12, 12)
    for (dims, B, heads) in [((16, 24, 12), 1, 2), ((16, 12, 12), 2, 1)]:
        D, H, W = dims
        for shifted in (False, True):
            ws, ss = SI.get_window_size(dims, win, (4, 6, 6) if shifted else (0, 0, 0))
            m, (Dp, Hp, Wp) = SI.window_map(D, H, W, ws, ss)
            N = ws[0] * ws[1] * ws[2]
            nW = m.size // N
            reg = SI.region_ids(Dp, Hp, Wp, ws, ss)
            rc, rc0 = SI.rc_codes(N, win)
            C_ = heads * 32
            nseq = B * nW
            qkv = rnd(nseq * N, 3 * C_, scale=1.0)
            table = (torch.randn(15 * 23 * 23, heads, device=dev) * 0.5)
            rc_t = torch.from_numpy(rc).to(dev)
            reg_t = torch.from_numpy(reg).to(dev) if reg is not None else None
            sscale = (torch.rand(B, device=dev) + 0.5)
            kw = dict(q_off=0, k_off=C_, v_off=2 * C_, bias_table=table, rc=rc_t, rc0=rc0, region=reg_t, n_win=nW, seq_scale=sscale, seqs_per_scale=nW)
            out, lse = K.attention_fwd(qkv, nseq, N, heads, 32, 0, 32 ** -0.5, **kw)
            qf = qkv.float().requires_grad_(True)
            tf = table.clone().requires_grad_(True)
            x = qf.view(nseq, N, 3, heads, 32).permute(2, 0, 3, 1, 4)
            idx = (rc_t[:, None] - rc_t[None, :] + rc0).long()
            bias = tf[idx.reshape(-1)].view(N, N, heads).permute(2, 0, 1)[None]
            if reg is not None:
                mk = torch.where(reg_t[:, :, None] != reg_t[:, None, :], -100.0, 0.0)
                bias = bias + mk.repeat(B, 1, 1)[:, None]
            o = attn_ref(x[0], x[1], x[2], bias)
            sc = sscale.repeat_interleave(nW)[:, None, None, None]
            ref = (o * sc).transpose(1, 2).reshape(nseq * N, C_)
            tag = f"stream win attn {dims} N={N} shifted={shifted}"
            rep(tag + " fwd", out, ref)
            dout = rnd(nseq * N, C_)
            ref.backward(dout.float())
            dtab = torch.zeros_like(table)
            dqkv = K.attention_bwd(dout, qkv, out, lse, nseq, N, heads, 32, 0, 32 ** -0.5, dbias_table=dtab, **kw)
            gq = qf.grad.clone()
            gq[:, :C_] *= 32 ** -0.5
            rep(tag + " bwd dq", dqkv[:, :C_], gq[:, :C_])
            rep(tag + " bwd dk", dqkv[:, C_:2 * C_], gq[:, C_:2 * C_])
            rep(tag + " bwd dv", dqkv[:, 2 * C_:], gq[:, 2 * C_:])
            rep(tag + " bwd dtable", dtab, tf.grad)
    for (nseq, Lq, heads) in [(2, 2352, 2), (3, 600, 3)]:
        Hd = heads * 64
        qkv = rnd(nseq * Lq, 3 * Hd, scale=1.0)
        km = torch.ones(nseq, Lq, dtype=torch.uint8, device=dev)
        for s in range(nseq):
            km[s, Lq - 5 * (s + 1):] = 0
        out, lse = K.attention_fwd(qkv, nseq, Lq, heads, 64, 1, 0.125, q_off=0, k_off=Hd, v_off=2 * Hd, keymask=km)
        qf = qkv.float().requires_grad_(True)
        x = qf.view(nseq, Lq, 3, heads, 64).permute(2, 0, 3, 1, 4)
        bias = torch.where(km.bool(), 0.0, float("-inf"))[:, None, None, :]
        o = attn_ref(x[0] * 0.125, x[1], x[2], bias)
        ref = o.transpose(1, 2).reshape(nseq * Lq, Hd)
        tag = f"stream bert attn nseq={nseq} L={Lq} h={heads}"
        rep(tag + " fwd", out, ref)
        lse_ref = torch.logsumexp((x[0] * 0.125) @ x[1].transpose(-1, -2) + bias, -1)
        rep(tag + " lse", lse, lse_ref.detach(), tol=2e-3)
        dout = rnd(nseq * Lq, Hd)
        ref.backward(dout.float())
        dqkv = K.attention_bwd(dout, qkv, out, lse, nseq, Lq, heads, 64, 1, 0.125, q_off=0, k_off=Hd, v_off=2 * Hd, keymask=km)
        rep(tag + " bwd dq", dqkv[:, :Hd], qf.grad[:, :Hd])
        rep(tag + " bwd dk", dqkv[:, Hd:2 * Hd], qf.grad[:, Hd:2 * Hd])
        rep(tag + " bwd dv", dqkv[:, 2 * Hd:], qf.grad[:, 2 * Hd:])
    # forced streaming at the resident kernels' shapes: same dropout stream, same masks -> same results up to bf16 rounding
    nseq, Lq, heads = 3, 432, 4
    Hd = heads * 64
    qkv = rnd(nseq * Lq, 3 * Hd, scale=1.0)
    km = torch.ones(nseq, Lq, dtype=torch.uint8, device=dev)
    km[1, 410:] = 0
    dout = rnd(nseq * Lq, Hd)
    kw = dict(q_off=0, k_off=Hd, v_off=2 * Hd, keymask=km, dropout_p=0.1, seed=77, offset=12345)
    o0, l0 = K.attention_fwd(qkv, nseq, Lq, heads, 64, 1, 0.125, **kw)
    o1, l1 = K.attention_fwd(qkv, nseq, Lq, heads, 64, 1, 0.125, stream_min_len=1, **kw)
    rep("stream vs resident bert+dropout fwd", o1, o0, tol=1e-2)
    rep("stream vs resident bert+dropout lse", l1, l0, tol=1e-4)
    g0 = K.attention_bwd(dout, qkv, o0, l0, nseq, Lq, heads, 64, 1, 0.125, **kw)
    g1 = K.attention_bwd(dout, qkv, o0, l0, nseq, Lq, heads, 64, 1, 0.125, stream_min_len=1, **kw)
    rep("stream vs resident bert+dropout bwd", g1, g0, tol=1e-2)
    dims, win7, B, heads = (8, 14, 14), (8, 7, 7), 2, 4
    ws, ss = SI.get_window_size(dims, win7, (4, 3, 3))
    m, (Dp, Hp, Wp) = SI.window_map(*dims, ws, ss)
    N = ws[0] * ws[1] * ws[2]
    nW = m.size // N
    reg_t = torch.from_numpy(SI.region_ids(Dp, Hp, Wp, ws, ss)).to(dev)
    rc, rc0 = SI.rc_codes(N, win7)
    rc_t = torch.from_numpy(rc).to(dev)
    C_ = heads * 32
    nseq = B * nW
    qkv = rnd(nseq * N, 3 * C_, scale=1.0)
    table = (torch.randn(15 * 13 * 13, heads, device=dev) * 0.5)
    dout = rnd(nseq * N, C_)
    kw = dict(q_off=0, k_off=C_, v_off=2 * C_, bias_table=table, rc=rc_t, rc0=rc0, region=reg_t, n_win=nW)
    o0, l0 = K.attention_fwd(qkv, nseq, N, heads, 32, 0, 32 ** -0.5, **kw)
    o1, l1 = K.attention_fwd(qkv, nseq, N, heads, 32, 0, 32 ** -0.5, stream_min_len=1, **kw)
    rep("stream vs resident window fwd", o1, o0, tol=1e-2)
    rep("stream vs resident window lse", l1, l0, tol=1e-3)      # the resident window kernels carry bias+mask as packed bf16
    t0, t1 = torch.zeros_like(table), torch.zeros_like(table)
    g0 = K.attention_bwd(dout, qkv, o0, l0, nseq, N, heads, 32, 0, 32 ** -0.5, dbias_table=t0, **kw)
    g1 = K.attention_bwd(dout, qkv, o0, l0, nseq, N, heads, 32, 0, 32 ** -0.5, dbias_table=t1, stream_min_len=1, **kw)
    rep("stream vs resident window bwd", g1, g0, tol=1e-2)
    rep("stream vs resident window dtable", t1, t0, tol=1e-2)


def check_attn_seq2seq():
    """mode-1 attention with the seq2seq mask (model.py:191-199; smtm pass): visual keys for everyone, text keys lower-triangular for
    text queries only (no padding mask there), with attention dropout off and on (mask recovered through the resident kernel)."""
    for (nseq, Lq, Lv, heads) in [(3, 232, 200, 4), (2, 432, 400, 2)]:
        Hd = heads * 64
        qkv = rnd(nseq * Lq, 3 * Hd, scale=1.0)
        km = torch.ones(nseq, Lq, dtype=torch.uint8, device=dev)
        km[:, Lq - 7:] = 0                                     # text padding: ignored inside the triangle, as in the reference
        kw = dict(q_off=0, k_off=Hd, v_off=2 * Hd, keymask=km, causal_from=Lv)
        out, lse = K.attention_fwd(qkv, nseq, Lq, heads, 64, 1, 0.125, **kw)
        allow = torch.zeros(Lq, Lq, dtype=torch.bool, device=dev)
        allow[:, :Lv] = True
        allow[Lv:, Lv:] = torch.tril(torch.ones(Lq - Lv, Lq - Lv, dtype=torch.bool, device=dev))
        bias = torch.where(allow, 0.0, float("-inf"))[None, None]
        qf = qkv.float().requires_grad_(True)
        x = qf.view(nseq, Lq, 3, heads, 64).permute(2, 0, 3, 1, 4)
        o = attn_ref(x[0] * 0.125, x[1], x[2], bias)
        ref = o.transpose(1, 2).reshape(nseq * Lq, Hd)
        tag = f"seq2seq attn nseq={nseq} L={Lq} Lv={Lv}"
        rep(tag + " fwd", out, ref)
        dout = rnd(nseq * Lq, Hd)
        ref.backward(dout.float())
        dqkv = K.attention_bwd(dout, qkv, out, lse, nseq, Lq, heads, 64, 1, 0.125, **kw)
        rep(tag + " bwd dq", dqkv[:, :Hd], qf.grad[:, :Hd])
        rep(tag + " bwd dk", dqkv[:, Hd:2 * Hd], qf.grad[:, Hd:2 * Hd])
        rep(tag + " bwd dv", dqkv[:, 2 * Hd:], qf.grad[:, 2 * Hd:])


def check_attn_colsum():
    """get_att's attention column sums (main_pretrain.py:211-215) from the forward kernel: sum over queries of the head-averaged
    probabilities, accumulated over calls (layers), against fp32 PyTorch; the attention output itself must not change."""
    for (nseq, Lq, heads) in [(3, 232, 4), (2, 432, 12)]:
        Hd = heads * 64
        km = torch.ones(nseq, Lq, dtype=torch.uint8, device=dev)
        km[1, Lq - 9:] = 0
        cs = torch.zeros(nseq, Lq, device=dev)
        ref_cs = torch.zeros(nseq, Lq, device=dev)
        for layer in range(2):
            qkv = rnd(nseq * Lq, 3 * Hd, scale=1.0)
            out0, _ = K.attention_fwd(qkv, nseq, Lq, heads, 64, 1, 0.125, q_off=0, k_off=Hd, v_off=2 * Hd, keymask=km)
            out1, _ = K.attention_fwd(qkv, nseq, Lq, heads, 64, 1, 0.125, q_off=0, k_off=Hd, v_off=2 * Hd, keymask=km, att_colsum=cs)
            x = qkv.float().view(nseq, Lq, 3, heads, 64).permute(2, 0, 3, 1, 4)
            bias = torch.where(km.bool(), 0.0, float("-inf"))[:, None, None, :]
            P = ((x[0] * 0.125) @ x[1].transpose(-1, -2) + bias).softmax(-1)
            ref_cs += P.mean(dim=1).sum(dim=1)
            rep(f"colsum pass leaves the output alone L={Lq} layer {layer}", out1, out0, tol=1e-6)
        rep(f"attention column sums nseq={nseq} L={Lq} h={heads} (2 layers)", cs, ref_cs, tol=5e-3)
    # with dropout the expectation is unchanged: sum over keys = number of queries per layer (within a few %)
    nseq, Lq, heads = 2, 432, 12
    Hd = heads * 64
    qkv = rnd(nseq * Lq, 3 * Hd, scale=1.0)
    cs = torch.zeros(nseq, Lq, device=dev)
    K.attention_fwd(qkv, nseq, Lq, heads, 64, 1, 0.125, q_off=0, k_off=Hd, v_off=2 * Hd, dropout_p=0.1, seed=5, offset=77, att_colsum=cs)
    tot = cs.sum(1)
    rep("column sums under dropout: total mass ~ L", tot, torch.full_like(tot, float(Lq)), tol=2e-2)


def check_gemm_fp8():
    """fp8 (OCP e4m3) build of the persistent GEMM (config 5's forward Linear layers): exact against fp32 PyTorch on the SAME
    quantised operands (the kernel's arithmetic), then the end-to-end quantisation error against the unquantised product, and the
    rate at the fusion FFN shape next to the bf16 build."""
    f8 = torch.float8_e4m3fn
    for (M, N, Kd, act) in [(512, 256, 128, 0), (1000, 1544, 768, 1), (4096, 3072, 768, 2)]:
        A = rnd(M, Kd, scale=1.0)
        W = rnd(N, Kd, scale=0.02)
        bias = torch.randn(N, device=dev) * 0.1
        res = rnd(M, N, scale=0.5)
        sa, sw = 16.0, 2048.0                                  # powers of two: |A| <~ 5 -> 80, |W| <~ 0.1 -> 200 (e4m3 max 448)
        A8, W8 = K.cast_fp8(A, sa), K.cast_fp8(W, sw)
        # the cast itself against torch's e4m3fn rounding
        rep(f"cast bf16->fp8 {M}x{Kd}", A8.view(f8).float(), (A.float() * sa).clamp(-448, 448).to(f8).float(), tol=1e-6)
        Aq, Wq = A8.view(f8).float() / sa, W8.view(f8).float() / sw
        pre = torch.empty((M, N), device=dev, dtype=BF) if act == 1 else None
        out = K.gemm(A8, W8, bias=bias, act=act, out_preact=pre, resid=res, fp8=True, alpha=1.0 / (sa * sw))
        z = Aq @ Wq.t() + bias
        ref = (torch.nn.functional.gelu(z) if act == 1 else (z.clamp_min(0) if act == 2 else z)) + res.float()
        rep(f"fp8 gemm {M}x{N}x{Kd} act={act} vs fp32 on the quantised operands", out, ref, tol=1e-2)
        if act == 1:
            rep(f"fp8 gemm {M}x{N}x{Kd} saved pre-activation", pre, z, tol=1e-2)
            code = torch.empty((M, N), device=dev, dtype=torch.uint8)          # the same call saving 8-bit GELU' codes instead
            out_c = K.gemm(A8, W8, bias=bias, act=1, out_preact=code, resid=res, fp8=True, alpha=1.0 / (sa * sw), code8=True)
            rep(f"fp8 gemm {M}x{N}x{Kd} act=1 (code8 build)", out_c, ref, tol=1e-2)
            zf = z.clone().requires_grad_(True)
            torch.nn.functional.gelu(zf).sum().backward()
            rep(f"fp8 gemm {M}x{N}x{Kd} code8 decode vs GELU'", code.float() * (1.26 / 255.0) - 0.13, zf.grad, tol=0.0045 / 1.13)
        z0 = A.float() @ W.float().t() + bias
        ref0 = (torch.nn.functional.gelu(z0) if act == 1 else (z0.clamp_min(0) if act == 2 else z0)) + res.float()
        err = (out.float() - ref0).norm() / ref0.norm()
        print(f"     end-to-end relative error of the fp8 path vs the unquantised product: {float(err):.3e}")
        rep(f"fp8 gemm {M}x{N}x{Kd} quantisation error (Frobenius, <= 4e-2)", (out.float() - ref0).norm().view(1) / ref0.norm(), torch.zeros(1, device=dev) + 1e-9, tol=4e7)
    # the 256x256 ping-pong main loop on fp8 operands (variant 7 forces it): plain, bias + q-scale + row scale, bias + GELU + saved
    # pre-activation, bias + residual + dropout(p = 0 here: exact), ragged M
    for (M, N, Kd) in [(1024, 512, 256), (3000, 768, 1024), (4096, 3072, 768)]:
        A = rnd(M, Kd, scale=1.0)
        W = rnd(N, Kd, scale=0.02)
        bias = torch.randn(N, device=dev) * 0.1
        res = rnd(M, N, scale=0.5)
        rs = torch.rand((M + 255) // 256, device=dev) + 0.5
        sa, sw = 16.0, 2048.0
        A8, W8 = K.cast_fp8(A, sa), K.cast_fp8(W, sw)
        Aq, Wq = A8.view(f8).float() / sa, W8.view(f8).float() / sw
        z = Aq @ Wq.t()
        al = 1.0 / (sa * sw)
        rep(f"fp8 pp {M}x{N}x{Kd} plain", K.gemm(A8, W8, fp8=True, alpha=al, variant=7), z, tol=1e-2)
        zq = z + bias
        zq[:, :N // 2] *= 0.125
        rep(f"fp8 pp {M}x{N}x{Kd} bias+colscale+rowscale", K.gemm(A8, W8, bias=bias, col_scale=0.125, col_scale_n=N // 2, row_scale=rs, rows_per_scale=256, fp8=True, alpha=al, variant=7),
            zq * rs.repeat_interleave(256)[:M, None], tol=1e-2)
        pre = torch.empty((M, N), device=dev, dtype=BF)
        rep(f"fp8 pp {M}x{N}x{Kd} bias+gelu", K.gemm(A8, W8, bias=bias, act=1, out_preact=pre, fp8=True, alpha=al, variant=7), torch.nn.functional.gelu(z + bias), tol=1e-2)
        rep(f"fp8 pp {M}x{N}x{Kd} saved pre-activation", pre, z + bias, tol=1e-2)
        rep(f"fp8 pp {M}x{N}x{Kd} bias+resid", K.gemm(A8, W8, bias=bias, resid=res, fp8=True, alpha=al, variant=7), z + bias + res.float(), tol=1e-2)
    for (M, N, Kk) in [(55296, 768, 3072), (36864, 1536, 6144)]:
        A = rnd(M, Kk, scale=1.0); W = rnd(N, Kk, scale=0.02)
        A8, W8 = K.cast_fp8(A, 16.0), K.cast_fp8(W, 2048.0)
        bias = torch.randn(N, device=dev) * 0.1
        res = rnd(M, N, scale=0.5)
        for name, fn in (("bf16 bias+resid", lambda: K.gemm(A, W, bias=bias, resid=res)), ("fp8 128^2 bias+resid", lambda: K.gemm(A8, W8, bias=bias, resid=res, fp8=True, alpha=1.0 / 32768.0, variant=6)),
                         ("fp8 ping-pong bias+resid", lambda: K.gemm(A8, W8, bias=bias, resid=res, fp8=True, alpha=1.0 / 32768.0, variant=7))):
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            print(f"     {name:26s} {M}x{N}x{Kk}: {ms:.3f} ms  {2.0 * M * N * Kk / ms / 1e9:.0f} TFLOP/s")
    M, N, Kk = 69120, 3072, 768
    A = rnd(M, Kk, scale=1.0); W = rnd(N, Kk, scale=0.02)
    A8, W8 = K.cast_fp8(A, 16.0), K.cast_fp8(W, 2048.0)
    bias = torch.randn(N, device=dev) * 0.1
    pre = torch.empty((M, N), device=dev, dtype=BF)
    for name, fn in (("bf16 plain", lambda: K.gemm(A, W)), ("fp8 plain", lambda: K.gemm(A8, W8, fp8=True, alpha=1.0 / 32768.0)),
                     ("bf16 bias+GELU+pre", lambda: K.gemm(A, W, bias=bias, act=1, out_preact=pre)),
                     ("fp8 bias+GELU+pre", lambda: K.gemm(A8, W8, bias=bias, act=1, out_preact=pre, fp8=True, alpha=1.0 / 32768.0)),
                     ("cast bf16->fp8 of A", lambda: K.cast_fp8(A, 16.0))):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"     {name:22s} {M}x{N}x{Kk}: {ms:.3f} ms" + (f"  {2.0 * M * N * Kk / ms / 1e9:.0f} TFLOP/s" if "cast" not in name else f"  {M * Kk * 3 / ms / 1e6:.0f} GB/s"))


# ------------------------------------------------------------------ misc
def check_misc():
    # VTM pair-score cross entropy + its f32 gradient (vmvm_vtm_ce) against torch in fp32: B clips x O candidates, class 0 = the positive
    for (B_, O_) in [(32, 4), (5, 3), (700, 4)]:
        lg = (torch.randn(B_, O_, device=dev) * 3.0).requires_grad_(True)
        ls = torch.nn.functional.cross_entropy(lg, torch.zeros(B_, dtype=torch.int64, device=dev))
        ls.backward()
        acc = torch.full((1,), 0.25, device=dev)
        dlg = K.vtm_ce(lg.detach().reshape(-1).contiguous(), B_, O_, acc)
        rep(f"vtm_ce loss B={B_} O={O_} (accumulated onto 0.25)", acc, ls.detach().view(1) + 0.25, tol=1e-5)
        rep(f"vtm_ce dlogits B={B_} O={O_}", dlg.view(B_, O_), lg.grad, tol=1e-5)
    B, T, H, W = 2, 4, 64, 96
    img = torch.randn(B, T, 3, H, W, device=dev)
    cols = K.patch_im2col(img)
    x = torch.nn.functional.pad(img.transpose(1, 2), (0, 0, 0, 0, 0, 1))                    # (B,3,T+1,H,W)
    w = torch.randn(32, 3, 2, 4, 4, device=dev)
    ref = torch.nn.functional.conv3d(x.to(BF).float(), w, stride=(1, 4, 4)).permute(0, 2, 3, 4, 1).reshape(-1, 32)
    rep("im2col hi+lo (via conv3d)", (cols[:, :96].float() + cols[:, 96:].float()) @ w.view(32, 96).t(), torch.nn.functional.conv3d(x, w, stride=(1, 4, 4)).permute(0, 2, 3, 4, 1).reshape(-1, 32), tol=1e-4)
    # fused PatchEmbed3D forward (conv + bias + LayerNorm, cover + zero frame applied while reading) against conv3d + layer_norm in fp32
    for (B_, T_, H_, W_, E_) in [(2, 4, 64, 96, 128), (1, 3, 32, 32, 96), (2, 2, 64, 32, 192)]:
        im = torch.randn(B_, T_, 3, H_, W_, device=dev)
        cv = (torch.rand(B_, T_, H_ // 32, W_ // 32, device=dev) < 0.3).to(torch.uint8)
        wq = (torch.randn(E_, 3, 2, 4, 4, device=dev) * 0.1).to(BF)
        bq, gq, beq = torch.randn(E_, device=dev) * 0.1, 1 + 0.1 * torch.randn(E_, device=dev), 0.1 * torch.randn(E_, device=dev)
        for cov_ in (None, cv):
            xm = im if cov_ is None else im * (1 - cov_.float().repeat_interleave(32, 2).repeat_interleave(32, 3)).unsqueeze(2)
            xin = torch.nn.functional.pad(xm.transpose(1, 2), (0, 0, 0, 0, 0, 1))
            zr = torch.nn.functional.conv3d(xin, wq.float(), bq, stride=(1, 4, 4)).permute(0, 2, 3, 4, 1).reshape(-1, E_)
            xr = torch.nn.functional.layer_norm(zr, (E_,), gq, beq, 1e-5)
            xo, zo, mo, ro = K.patch_embed_fwd(im, cov_, wq.view(E_, 96).contiguous(), bq, gq, beq, 1e-5)
            tag = f"patch_embed_fwd B={B_} T={T_} {H_}x{W_} E={E_} cov={int(cov_ is not None)}"
            rep(tag + " z", zo, zr, tol=2e-4)
            rep(tag + " x", xo, xr, tol=8e-3)
            rep(tag + " mean", mo, zr.mean(1), tol=2e-4)
            rep(tag + " rstd", ro, (zr.var(1, unbiased=False) + 1e-5).rsqrt(), tol=2e-4)
    # encvideo assemble
    hw, Hd = 6, 64
    fc = rnd(B * T * hw, Hd)
    cls, pos, ln = torch.randn(Hd, device=dev), torch.randn(1 + 14 * 14, Hd, device=dev), torch.randn(6, Hd, device=dev)
    out = K.encvideo_assemble(fc, cls, pos, ln, B, T, hw, Hd)
    ref = torch.cat([cls.expand(B, T, 1, Hd), fc.float().view(B, T, hw, Hd)], 2) + pos[None, None, :1 + hw] + ln[None, :T, None]
    rep("encvideo assemble", out, ref.reshape(-1, Hd))
    dpre = rnd(B * T * (1 + hw), Hd)
    dcls, dpos, dlen = torch.zeros(Hd, device=dev), torch.zeros_like(pos), torch.zeros_like(ln)
    dfc = K.encvideo_assemble_bwd(dpre, dcls, dpos, dlen, B, T, hw, Hd)
    d4 = dpre.float().view(B, T, 1 + hw, Hd)
    rep("encvideo bwd dfc", dfc, d4[:, :, 1:].reshape(-1, Hd), tol=0)
    rep("encvideo bwd dcls", dcls, d4[:, :, 0].sum((0, 1)), tol=1e-3)
    rep("encvideo bwd dpos", dpos[:1 + hw], d4.sum((0, 1)), tol=1e-3)
    rep("encvideo bwd dlen", dlen[:T], d4.sum((0, 2)), tol=1e-3)
    # bert embed
    X, V = 32, 1000
    txt = torch.randint(0, V, (B, X), device=dev)
    word, pemb, temb = torch.randn(V, Hd, device=dev), torch.randn(512, Hd, device=dev), torch.randn(2, Hd, device=dev)
    out = K.bert_embed(txt, word, pemb, temb[0])
    rep("bert embed", out, (word[txt] + pemb[:X][None] + temb[0]).reshape(-1, Hd))
    ds = rnd(B * X, Hd)
    dword, dposb, dtype0 = torch.zeros_like(word), torch.zeros_like(pemb), torch.zeros(Hd, device=dev)
    K.bert_embed_bwd(txt, ds, dword, dposb, dtype0)
    refw = torch.zeros_like(word).index_add_(0, txt.reshape(-1), ds.float())
    rep("bert embed bwd dword", dword, refw, tol=1e-3)
    rep("bert embed bwd dpos", dposb[:X], ds.float().view(B, X, Hd).sum(0), tol=1e-3)
    rep("bert embed bwd dtype", dtype0, ds.float().sum(0), tol=1e-3)
    # cross entropy
    M, V, ld = 64, 30522, 30528
    logits = torch.randn(M, ld, device=dev) * 3
    tgt = torch.randint(0, V, (M,), device=dev)
    tgt[::3] = -1
    loss = torch.zeros(1, device=dev)
    dlog = K.cross_entropy(logits, V, tgt, loss)
    lf = logits[:, :V].clone().requires_grad_(True)
    refl = torch.nn.functional.cross_entropy(lf, tgt, ignore_index=-1)
    refl.backward()
    rep("cross entropy loss", loss, refl.detach().view(1), tol=1e-4)
    rep("cross entropy dlogits", dlog[:, :V], lf.grad, tol=1e-2)
    rep("cross entropy dlogits pad", dlog[:, V:], torch.zeros(M, ld - V, device=dev), tol=0)
    # pixel l1
    B, T, h, w, ps = 2, 2, 3, 2, 32
    pred = rnd(B * T * h * w, 3 * ps * ps)
    tgt_img = torch.randn(B, T, 3, h * ps, w * ps, device=dev)
    cov = (torch.rand(B, T, h, w, device=dev) < 0.4).to(torch.uint8)
    cov[0, 0, 0, 0] = 1
    full = cov.float()[:, :, None, :, None, :, None].expand(-1, -1, 3, -1, ps, -1, ps).reshape(B, T, 3, h * ps, w * ps)
    msum = full.sum().view(1)
    loss = torch.zeros(1, device=dev)
    dpred = K.pixel_l1(pred, tgt_img, cov.view(-1), msum, loss, B, T, h, w, ps)
    pf = pred.float().requires_grad_(True)
    xx = pf.view(B * T, h, w, 3 * ps * ps).permute(0, 3, 1, 2)
    xx = torch.nn.functional.pixel_shuffle(xx, ps).view(B, T, 3, h * ps, w * ps)
    refl = ((xx - tgt_img).abs() * full).sum() / (full.sum() + 1e-5) / 3
    refl.backward()
    rep("pixel l1 loss", loss, refl.detach().view(1), tol=1e-4)
    rep("pixel l1 dpred", dpred, pf.grad, tol=1e-2)
    # rowdot
    M, Kd = 12, 1536
    hid = rnd(M, Kd)
    wv, bv = torch.randn(Kd, device=dev) * 0.05, torch.randn(1, device=dev)
    o = K.rowdot(hid, wv, bv, 20.0)
    rep("rowdot", o, (hid.float() @ wv + bv) * 20.0, tol=1e-3)
    do = torch.randn(M, device=dev)
    dw, dbv = torch.zeros(Kd, device=dev), torch.zeros(1, device=dev)
    dh = K.rowdot_bwd(hid, wv, do, 20.0, dw, dbv)
    rep("rowdot bwd dhid", dh, (do * 20.0)[:, None] * wv[None])
    rep("rowdot bwd dw", dw, (do * 20.0) @ hid.float(), tol=1e-3)
    rep("rowdot bwd db", dbv, (do * 20.0).sum().view(1), tol=1e-3)
    # colsum
    Xc = rnd(5000, 264)
    o = torch.zeros(264, device=dev)
    rs = torch.rand(5, device=dev)
    K.colsum(Xc, o, rs, 1000)
    rep("colsum row-scaled", o, (Xc.float() * rs.repeat_interleave(1000)[:, None]).sum(0), tol=1e-3)
    # batched transpose
    src = rnd(4096 + 200 * 136 + 64)
    dstT = torch.zeros_like(src)
    ents = []
    for (o, N_, K_) in ((0, 64, 64), (4096, 200, 136)):
        for tr in range(-(-N_ // 64)):
            for tc in range(-(-K_ // 64)):
                ents.append((o, N_, K_, (tr << 16) | tc))
    K.transpose_batched(src, dstT, torch.tensor(ents, dtype=torch.int32, device=dev))
    rep("transpose_batched 64x64", dstT[:4096].view(64, 64), src[:4096].view(64, 64).t(), tol=0)
    rep("transpose_batched 200x136", dstT[4096:4096 + 200 * 136].view(136, 200), src[4096:4096 + 200 * 136].view(200, 136).t(), tol=0)
    # (round 6: the kernel is a resident grid of 2 048 workgroups walking the tile table through two LDS buffers: MORE tiles than workgroups)
    Nb, Kb = 3000, 3016
    srcb = rnd(Nb * Kb + 1000 * 72)
    dstb = torch.zeros_like(srcb)
    ents = [(0, Nb, Kb, (tr << 16) | tc) for tr in range(-(-Nb // 64)) for tc in range(-(-Kb // 64))]
    ents += [(Nb * Kb, 1000, 72, (tr << 16) | tc) for tr in range(-(-1000 // 64)) for tc in range(2)]
    assert len(ents) > 2048 * 1.05
    K.transpose_batched(srcb, dstb, torch.tensor(ents, dtype=torch.int32, device=dev))
    rep("transpose_batched 3000x3016 (tiles > resident grid)", dstb[:Nb * Kb].view(Kb, Nb), srcb[:Nb * Kb].view(Nb, Kb).t(), tol=0)
    rep("transpose_batched 1000x72 behind it", dstb[Nb * Kb:].view(72, 1000), srcb[Nb * Kb:].view(1000, 72).t(), tol=0)
    # optimizer
    n = 100003
    p, g = torch.randn(n, device=dev), torch.randn(n, device=dev) * 3
    m, v = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    pb = torch.zeros(n, device=dev, dtype=BF)
    ss = torch.zeros(1, device=dev)
    K.sumsq(g, ss)
    rep("sumsq", ss, (g.double() ** 2).sum().float().view(1), tol=1e-4)
    pr = p.clone().requires_grad_(True)
    opt = torch.optim.AdamW([pr], lr=1e-2, betas=(0.9, 0.98), weight_decay=1e-3)
    for step in (1, 2):
        pr.grad = g.clone()
        torch.nn.utils.clip_grad_norm_([pr], 1.0)
        opt.step()
        K.adamw(p, g, m, v, pb, lr=1e-2, weight_decay=1e-3, beta1=0.9, beta2=0.98, eps=1e-8, step=step, sumsq_t=ss, max_grad_norm=1.0)
    rep("adamw 2 steps (clip)", p, pr.detach(), tol=1e-5)
    rep("adamw bf16 shadow", pb, p, tol=1e-2)


def bench_gemm():
    global WS
    WS = torch.empty(256 << 20, device=dev, dtype=torch.uint8)
    print("---- gemm timing (ms, TFLOP/s) vs torch.matmul (hipBLASLt ceiling)")
    for (M, N, K_) in [(8192, 8192, 8192), (69120, 3072, 768), (69120, 768, 3072), (802816, 384, 128), (50176, 2048, 512)]:
        A, B = rnd(M, K_), rnd(N, K_)
        for name, fn in (("vmvm", lambda: K.gemm(A, B)), ("v128", lambda: K.gemm(A, B, variant=3)), ("v256", lambda: K.gemm(A, B, variant=4)), ("p3", lambda: K.gemm(A, B, variant=5)), ("torch", lambda: A @ B.t())):
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fn()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            print(f"     {name:6s} {M}x{N}x{K_}: {ms:.3f} ms  {2.0 * M * N * K_ / ms / 1e9:.1f} TF")
        Bt = B.t().contiguous()
        At = A.t().contiguous()
        dy = rnd(M, N)
        gw = torch.zeros(N, K_, device=dev)
        for name, fn in (("vmvm NN", lambda: K.gemm(A, Bt, b_kmajor=False)), ("p3 NN", lambda: K.gemm(A, Bt, b_kmajor=False, variant=5)), ("vmvm TN", lambda: K.gemm(At, Bt, a_kmajor=False, b_kmajor=False)), ("p3 TN", lambda: K.gemm(At, Bt, a_kmajor=False, b_kmajor=False, variant=5)),
                         ("p3 wgrad ws", lambda: K.gemm(dy, A, a_kmajor=False, b_kmajor=False, M=N, N=K_, K=M, out=gw, accumulate=True, workspace=WS, variant=5)),
                         ("wgrad", lambda: K.gemm(dy, A, a_kmajor=False, b_kmajor=False, M=N, N=K_, K=M, out=gw, accumulate=True)),
                         ("wgrad ws", lambda: K.gemm(dy, A, a_kmajor=False, b_kmajor=False, M=N, N=K_, K=M, out=gw, accumulate=True, workspace=WS))):
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fn()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            print(f"     {name:6s} {M}x{N}x{K_}: {ms:.3f} ms  {2.0 * M * N * K_ / ms / 1e9:.1f} TF")


def bench_attn():
    print("---- attention timing")
    # the four Swin-B stages of the C2 workload (8 frames, each doubled -> D=8 -> window (8,7,7) = 392 tokens) + a half-depth window
    cases = [("stage-1", 32, 4, 392, (8, 7, 7), (8, 56, 56), True), ("stage-2", 32, 8, 392, (8, 7, 7), (8, 28, 28), True),
             ("stage-3", 32, 16, 392, (8, 7, 7), (8, 14, 14), True), ("stage-3 unshifted", 32, 16, 392, (8, 7, 7), (8, 14, 14), False),
             ("stage-4", 32, 32, 392, (8, 7, 7), (8, 7, 7), False), ("stage-3/D=4", 32, 16, 196, (4, 7, 7), (4, 14, 14), True)]
    layouts = [int(x) for x in os.environ.get("VMVM_BENCH_LAYOUTS", "0,1").split(",")]
    only = os.environ.get("VMVM_BENCH_ONLY")
    if only:
        cases = [c for c in cases if c[0] in only.split(",")]
    for (label, B, heads, N, ws, dims, sh), layout in [(c, l) for c in cases for l in layouts]:
        shift = (0, 3, 3) if sh else (0, 0, 0)
        if layout and not SI.win3_ok(ws, shift):
            continue
        label = label + (" layout=1" if layout else "")
        nW = (dims[1] // 7) * (dims[2] // 7)
        nseq = B * nW
        C_ = heads * 32
        qkv = rnd(nseq * N, 3 * C_)
        rc, rc0 = SI.rc_codes(N, (8, 7, 7))
        reg_np = SI.region_ids(dims[0], dims[1], dims[2], ws, shift) if any(shift) else None
        if layout:
            pm = SI.win3_perm()
            rc = np.ascontiguousarray(rc[pm])
            reg_np = None if reg_np is None else np.ascontiguousarray(reg_np[:, pm])
        rc_t = torch.from_numpy(rc).to(dev)
        table = torch.randn(2535, heads, device=dev) * 0.1
        reg = torch.from_numpy(reg_np).to(dev) if reg_np is not None else None
        f = lambda: K.attention_fwd(qkv, nseq, N, heads, 32, 0, 32 ** -0.5, q_off=0, k_off=C_, v_off=2 * C_, bias_table=table, rc=rc_t, rc0=rc0, region=reg, n_win=nW, win_layout=layout)
        out, lse = f()
        dout = rnd(nseq * N, C_)
        dtab = torch.zeros_like(table)
        b = lambda: K.attention_bwd(dout, qkv, out, lse, nseq, N, heads, 32, 0, 32 ** -0.5, q_off=0, k_off=C_, v_off=2 * C_, bias_table=table, rc=rc_t, rc0=rc0, region=reg, n_win=nW, dbias_table=dtab, win_layout=layout)
        fl = 4.0 * nseq * heads * N * N * 32
        b0 = lambda: K.attention_bwd(dout, qkv, out, lse, nseq, N, heads, 32, 0, 32 ** -0.5, q_off=0, k_off=C_, v_off=2 * C_, bias_table=table, rc=rc_t, rc0=rc0, region=reg, n_win=nW, dbias_table=None, win_layout=layout)
        for name, fn, mult in (("win fwd", f, 1), ("win bwd", b, 2.5), ("win bwd (no table grad)", b0, 2.5)):
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fn()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            print(f"     {name}: {ms:.3f} ms  {fl * mult / ms / 1e9:.1f} TF ({label} shape, B={B})")
        del qkv, out, lse, dout
    if os.environ.get("VMVM_BENCH_WINDOW_ONLY"):
        return
    nseq, Lq, heads = 160, 432, 12
    Hd = 768
    qkv = rnd(nseq * Lq, 3 * Hd)
    km = torch.ones(nseq, Lq, dtype=torch.uint8, device=dev)
    f = lambda: K.attention_fwd(qkv, nseq, Lq, heads, 64, 1, 0.125, q_off=0, k_off=Hd, v_off=2 * Hd, keymask=km, dropout_p=0.1, seed=1)
    out, lse = f()
    dout = rnd(nseq * Lq, Hd)
    b = lambda: K.attention_bwd(dout, qkv, out, lse, nseq, Lq, heads, 64, 1, 0.125, q_off=0, k_off=Hd, v_off=2 * Hd, keymask=km, dropout_p=0.1, seed=1)
    fl = 4.0 * nseq * heads * Lq * Lq * 64
    f0 = lambda: K.attention_fwd(qkv, nseq, Lq, heads, 64, 1, 0.125, q_off=0, k_off=Hd, v_off=2 * Hd, keymask=km, dropout_p=0.0, seed=1)
    b0 = lambda: K.attention_bwd(dout, qkv, out, lse, nseq, Lq, heads, 64, 1, 0.125, q_off=0, k_off=Hd, v_off=2 * Hd, keymask=km, dropout_p=0.0, seed=1)
    fs = lambda: K.attention_fwd(qkv, nseq, Lq, heads, 64, 1, 0.125, q_off=0, k_off=Hd, v_off=2 * Hd, keymask=km, dropout_p=0.1, seed=1, stream_min_len=1)
    bs = lambda: K.attention_bwd(dout, qkv, out, lse, nseq, Lq, heads, 64, 1, 0.125, q_off=0, k_off=Hd, v_off=2 * Hd, keymask=km, dropout_p=0.1, seed=1, stream_min_len=1)
    dm = K.attention_drop_mask(nseq, Lq, heads, 64, 1, 0.1, dev)
    fm = lambda: K.attention_fwd(qkv, nseq, Lq, heads, 64, 1, 0.125, q_off=0, k_off=Hd, v_off=2 * Hd, keymask=km, dropout_p=0.1, seed=1, drop_mask=dm)
    bm = lambda: K.attention_bwd(dout, qkv, out, lse, nseq, Lq, heads, 64, 1, 0.125, q_off=0, k_off=Hd, v_off=2 * Hd, keymask=km, dropout_p=0.1, seed=1, drop_mask=dm)
    for name, fn, mult in (("bert fwd", f, 1), ("bert bwd", b, 2.5), ("bert fwd (stores its dropout decisions)", fm, 1), ("bert bwd (reads them)", bm, 2.5),
                           ("bert fwd (no dropout)", f0, 1), ("bert bwd (no dropout)", b0, 2.5),
                           ("bert fwd (streaming kernels)", fs, 1), ("bert bwd (streaming kernels)", bs, 2.5)):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"     {name}: {ms:.3f} ms  {fl * mult / ms / 1e9:.1f} TF (B=32 x 5 seqs)")


def bench_attn_c5():
    """streaming attention kernels at the config-5 shapes (B = 4 clips): Swin-L stage-3 windows (8 per clip x 24 heads x 1152 tokens,
    head_dim 32, shifted) and the fusion sequences (B * 5 x 12 heads x 2352 tokens, head_dim 64, key mask + dropout)"""
    print("---- attention timing, config-5 shapes")
    B = 4
    heads, N, ws, dims = 24, 1152, (8, 12, 12), (16, 24, 24)
    nW = (dims[0] // 8) * (dims[1] // 12) * (dims[2] // 12)
    nseq = B * nW
    C_ = heads * 32
    qkv = rnd(nseq * N, 3 * C_)
    rc, rc0 = SI.rc_codes(N, ws)
    rc_t = torch.from_numpy(rc).to(dev)
    table = torch.randn(15 * 23 * 23, heads, device=dev) * 0.1
    reg = torch.from_numpy(SI.region_ids(dims[0], dims[1], dims[2], ws, (4, 6, 6))).to(dev)
    kw = dict(q_off=0, k_off=C_, v_off=2 * C_, bias_table=table, rc=rc_t, rc0=rc0, region=reg, n_win=nW)
    f = lambda: K.attention_fwd(qkv, nseq, N, heads, 32, 0, 32 ** -0.5, **kw)
    out, lse = f()
    dout = rnd(nseq * N, C_)
    dtab = torch.zeros_like(table)
    b = lambda: K.attention_bwd(dout, qkv, out, lse, nseq, N, heads, 32, 0, 32 ** -0.5, dbias_table=dtab, **kw)
    b0 = lambda: K.attention_bwd(dout, qkv, out, lse, nseq, N, heads, 32, 0, 32 ** -0.5, dbias_table=None, **kw)
    fl = 4.0 * nseq * heads * N * N * 32
    tests = [("c5 win fwd", f, 1, fl), ("c5 win bwd", b, 2.5, fl), ("c5 win bwd (no table grad)", b0, 2.5, fl)]
    nseq2, Lq, heads2, Hd = B * 5, 2352, 12, 768
    qkv2 = rnd(nseq2 * Lq, 3 * Hd)
    km = torch.ones(nseq2, Lq, dtype=torch.uint8, device=dev)
    f2 = lambda: K.attention_fwd(qkv2, nseq2, Lq, heads2, 64, 1, 0.125, q_off=0, k_off=Hd, v_off=2 * Hd, keymask=km, dropout_p=0.1, seed=1)
    out2, lse2 = f2()
    dout2 = rnd(nseq2 * Lq, Hd)
    b2 = lambda: K.attention_bwd(dout2, qkv2, out2, lse2, nseq2, Lq, heads2, 64, 1, 0.125, q_off=0, k_off=Hd, v_off=2 * Hd, keymask=km, dropout_p=0.1, seed=1)
    fl2 = 4.0 * nseq2 * heads2 * Lq * Lq * 64
    tests += [("c5 fusion fwd", f2, 1, fl2), ("c5 fusion bwd", b2, 2.5, fl2)]
    for name, fn, mult, flops in tests:
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"     {name}: {ms:.3f} ms  {flops * mult / ms / 1e9:.1f} TF (B={B})")


def bench_ln():
    print("---- layernorm timing (ms, algorithmic GB/s = bf16 tensors touched)")
    K.set_workspace(torch.empty(64 << 20, device=dev, dtype=torch.uint8))
    for (M, Cc) in [(69120, 768), (50176, 512), (200704, 256), (802816, 128), (12544, 1024)]:
        x, dy, add = rnd(M, Cc), rnd(M, Cc), rnd(M, Cc)
        g, b = torch.randn(Cc, device=dev), torch.randn(Cc, device=dev)
        y, mean, rstd = K.layernorm_fwd(x, g, b, 1e-5)
        dg, db = torch.zeros(Cc, device=dev), torch.zeros(Cc, device=dev)
        for name, fn, nt in (("ln fwd", lambda: K.layernorm_fwd(x, g, b, 1e-5), 2), ("ln bwd", lambda: K.layernorm_bwd(dy, x, g, mean, rstd, dg, db), 3),
                             ("ln bwd+add", lambda: K.layernorm_bwd(dy, x, g, mean, rstd, dg, db, dX_add=add), 4)):
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            print(f"     {name:10s} {M}x{Cc}: {ms:.3f} ms  {nt * M * Cc * 2 / ms / 1e6:.0f} GB/s")
    for (M, N) in [(55296, 3072), (55296, 768), (50176, 2048), (50176, 512), (802816, 512), (802816, 128)]:
        x = rnd(M, N)
        out = torch.zeros(N, device=dev)
        rs = torch.ones(M // 392 + 1, device=dev)
        for name, fn in (("colsum", lambda: K.colsum(x, out)), ("colsum rs", lambda: K.colsum(x, out, row_scale=rs, rows_per_scale=392))):
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            print(f"     {name:10s} {M}x{N}: {ms:.3f} ms  {M * N * 2 / ms / 1e6:.0f} GB/s")


if __name__ == "__main__":
    torch.manual_seed(0)
    which = sys.argv[1:] or ["probe", "gemm", "cs", "f16", "big", "epi", "ln", "lng", "attnw", "attnsp", "attnmb", "attnnf", "attnb", "attns", "attnc", "attna", "misc", "bench"]
    table = dict(probe=check_probe, gemm=check_gemm_layouts, cs=check_gemm_colsum, f16=check_gemm_fp16_conv, big=check_gemm_big, p3=check_gemm_p3, epi=check_gemm_epilogues, ln=check_ln, lng=check_ln_gather,
                 attnw=check_attn_window, attnsp=check_attn_window_spike, attnmb=check_attn_window_mask_boundary, attnnf=check_attn_window_nonfinite, attnb=check_attn_bert, attns=check_attn_stream, attnc=check_attn_seq2seq, attna=check_attn_colsum, f8=check_gemm_fp8, misc=check_misc, dvae=check_dvae_passes, pool=check_pool_grad)
    for w in which:
        if w == "bench":
            run(bench_gemm); run(bench_attn)
        elif w == "benchattn":
            run(bench_attn)
        elif w == "benchattn5":
            run(bench_attn_c5)
        elif w == "benchln":
            run(bench_ln)
        else:
            run(table[w])
    bad = [r for r in RESULTS if r[3]]
    print(f"==== {len(RESULTS) - len(bad)} ok, {len(bad)} FAILED")
    for r in bad:
        print("   FAILED:", r[0])
