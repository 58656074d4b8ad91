"""TSV container (utils/tsv_file.py, dataset.py:136-146) against outputs of the reference's own TSVFile / CompositeTSVFile /
create_lineidx / Dataset_Base.sampling on the same files (tests/golden/tsv.json, written by tools/gen_goldens.py --tsv-only)."""
import base64
import io
import json
import os

import numpy as np

from pytorch_empirical_mvm_amd import tsv as TSV

G = os.path.join(os.path.dirname(__file__), "golden")


def _write_files(root):
    """the same deterministic files tools/gen_goldens.py:write_tsv_fixture_files wrote for the reference"""
    names = ["img_a.tsv", "img_b.tsv"]
    for fi, name in enumerate(names):
        with open(os.path.join(root, name), "w") as f:
            for r in range(5 + 2 * fi):
                payload = [base64.b64encode(bytes((7 * r + 3 * c + fi + k) % 251 for k in range(40 + 17 * r + c))).decode() for c in range(1 + (r % 3))]
                key = f"vid{fi}_{r:03d}" + ("x" * (30 * (r % 2)))
                f.write("\t".join([key, f" caption {r} of file {fi} "] + payload) + "\n")
    with open(os.path.join(root, "files.txt"), "w") as f:
        f.write("\n".join(names) + "\n")
    with open(os.path.join(root, "seq.tsv"), "w") as f:
        for src, row in [(0, 2), (1, 6), (1, 0), (0, 4), (0, 0), (1, 3)]:
            f.write(f"{src}\t{row}\n")
    return names


def test_tsv_file_lineidx_rows_keys_and_composite_match_the_reference(tmp_path):
    ref = json.load(open(os.path.join(G, "tsv.json")))
    root = str(tmp_path)
    names = _write_files(root)
    for name in names:
        t = TSV.TSVFile(os.path.join(root, name), generate_lineidx=True)
        want = ref[name]
        assert [int(x) for x in open(t.lineidx).read().split()] == want["lineidx"]
        assert t.num_rows() == want["num_rows"] == len(t) == want["length"]
        assert [t.seek(i) for i in range(len(t))] == want["rows"]
        assert [t.seek_first_column(i) for i in range(len(t))] == want["keys"] == [t.get_key(i) for i in range(len(t))]
        assert t[len(t) - 1] == want["getitem_last"]
        assert not os.path.exists(t.lineidx + ".tmp")
        t2 = TSV.TSVFile(os.path.join(root, name))               # existing .lineidx is reused, random access in any order
        assert t2.seek(3) == want["rows"][3] and t2.seek(0) == want["rows"][0]
    c = TSV.CompositeTSVFile(os.path.join(root, "files.txt"), os.path.join(root, "seq.tsv"), root=root)
    want = ref["composite"]
    assert c.num_rows() == want["num_rows"] == len(c)
    assert [c[i] for i in range(len(c))] == want["rows"]
    assert [c.get_key(i) for i in range(len(c))] == want["keys"]
    assert c.get_composite_source_idx() == want["source_idx"]
    assert TSV.load_list_file(os.path.join(root, "files.txt")) == want["file_list"]
    c2 = TSV.CompositeTSVFile(names, os.path.join(root, "seq.tsv"), root=root)      # a python list instead of a list file
    assert [c2[i] for i in range(len(c2))] == want["rows"]


def test_frame_sampling_matches_the_reference_and_frames_decode():
    ref = json.load(open(os.path.join(G, "tsv.json")))["sampling"]
    for k, want in ref.items():
        a, b, n = (int(x) for x in k.split(","))
        assert TSV.sampling(a, b, n) == want, k
    from PIL import Image
    bufs = []
    for i, (w, h) in enumerate([(64, 48), (40, 80), (32, 32)]):
        a = (np.arange(w * h * 3, dtype=np.int64).reshape(h, w, 3) * (i + 3) % 256).astype(np.uint8)
        bio = io.BytesIO()
        Image.fromarray(a).save(bio, format="PNG")
        bufs.append(base64.b64encode(bio.getvalue()).decode())
    assert TSV.str2img(bufs[0]).size == (64, 48)
    for mode in ("img_center_crop", "pad_resize"):
        clip = TSV.frames_to_clip(bufs, 32, mode)
        assert tuple(clip.shape) == (3, 3, 32, 32) and bool(np.isfinite(clip.numpy()).all())
    # the 32x32 frame passes through both transforms unchanged: exact ImageNet normalisation of the pixels
    a = (np.arange(32 * 32 * 3, dtype=np.int64).reshape(32, 32, 3) * 5 % 256).astype(np.float32).transpose(2, 0, 1) / 255.0
    want = (a - np.array([0.485, 0.456, 0.406], np.float32).reshape(3, 1, 1)) / np.array([0.229, 0.224, 0.225], np.float32).reshape(3, 1, 1)
    np.testing.assert_allclose(TSV.frames_to_clip(bufs[2:], 32)[0].numpy(), want, rtol=1e-6, atol=1e-6)
