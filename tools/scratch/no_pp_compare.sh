# per-shape GEMM time inside a step: dispatcher's choice vs the 128x128 persistent kernel forced everywhere
python tools/gemm_shapes.py > gpurun_out/gs_old.txt 2>&1
VMVM_NO_PP=1 python tools/gemm_shapes.py > gpurun_out/gs_new.txt 2>&1
python - <<'PY'
def load(f):
    d = {}
    for l in open(f):
        t = l.split()
        if len(t) >= 8 and t[0].isdigit():
            d[(t[0], t[1], t[2], t[3], " ".join(t[8:]))] = (int(t[4]), float(t[5]), float(t[6]))
    return d
a, b = load("gpurun_out/gs_old.txt"), load("gpurun_out/gs_new.txt")
print("total dispatch %.2f ms  no-pp %.2f ms" % (sum(v[1] for v in a.values()), sum(v[1] for v in b.values())))
for k in sorted(a, key=lambda k: -a[k][1]):
    if k in b and abs(b[k][2] / a[k][2] - 1) > 0.03:
        print("%8s %6s %8s %s n=%3d  dispatch %7.1f us  no-pp %7.1f us  %+5.1f %%  %s" % (k[0], k[1], k[2], k[3], a[k][0], a[k][2], b[k][2], 100 * (b[k][2] / a[k][2] - 1), k[4]))
PY
