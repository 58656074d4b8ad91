"""TSV container of the reference's data pipeline (`utils/tsv_file.py:43-159`, `dataset.py:36-38,136-164`): one sample per
line, tab-separated columns (key, ..., base64-encoded frames), random access through a `.lineidx` side file of byte offsets; a
composite view over several files through a (source, row) sequence file.  Host-side format code, no GPU involved -- it sits on the
input side of the step (`img`, `txt`, `mask` batches) and keeps the reference's call surface (`TSVFile(path)[i]`, `num_rows`,
`seek`, `seek_first_column`, `get_key`, `CompositeTSVFile`, `create_lineidx`, `load_list_file`)."""
import base64
import io
import os
import os.path as op


def create_lineidx(filein, idxout):
    """utils/tsv_file.py:18-27: byte offset of every line, one per line, written atomically (tmp + rename)"""
    tmp = idxout + ".tmp"
    with open(filein, "rb") as fin, open(tmp, "w") as fout:
        size = os.fstat(fin.fileno()).st_size
        pos = 0
        while pos != size:
            fout.write(str(pos) + "\n")
            fin.readline()
            pos = fin.tell()
    os.rename(tmp, idxout)


def read_to_character(fp, c):
    """utils/tsv_file.py:30-40: read up to (not including) the first `c`, 32 characters at a time"""
    out = []
    while True:
        s = fp.read(32)
        assert s != ""
        if c in s:
            out.append(s[:s.index(c)])
            break
        out.append(s)
    return "".join(out)


def load_list_file(fname):
    """utils/tsv_file.py:151-157"""
    with open(fname, "r") as fp:
        lines = [ln.strip() for ln in fp.readlines()]
    if lines and lines[-1] == "":
        lines = lines[:-1]
    return lines


class TSVFile:
    """utils/tsv_file.py:43-113.  The file handle is re-opened after a fork (DataLoader workers), as the reference does."""

    def __init__(self, tsv_file, generate_lineidx=False):
        self.tsv_file = tsv_file
        self.lineidx = op.splitext(tsv_file)[0] + ".lineidx"
        self._fp = None
        self._lineidx = None
        self.pid = None
        if not op.isfile(self.lineidx) and generate_lineidx:
            create_lineidx(self.tsv_file, self.lineidx)

    def __del__(self):
        if self._fp:
            self._fp.close()

    def __repr__(self):
        return "TSVFile(tsv_file='{}')".format(self.tsv_file)

    __str__ = __repr__

    def num_rows(self):
        self._ensure_lineidx_loaded()
        return len(self._lineidx)

    __len__ = num_rows

    def seek(self, idx):
        self._ensure_tsv_opened()
        self._ensure_lineidx_loaded()
        self._fp.seek(self._lineidx[idx])
        return [s.strip() for s in self._fp.readline().split("\t")]

    __getitem__ = seek

    def seek_first_column(self, idx):
        self._ensure_tsv_opened()
        self._ensure_lineidx_loaded()
        self._fp.seek(self._lineidx[idx])
        return read_to_character(self._fp, "\t")

    get_key = seek_first_column

    def _ensure_lineidx_loaded(self):
        if self._lineidx is None:
            with open(self.lineidx, "r") as fp:
                self._lineidx = [int(i.strip()) for i in fp.readlines()]

    def _ensure_tsv_opened(self):
        if self._fp is None or self.pid != os.getpid():
            self._fp = open(self.tsv_file, "r")
            self.pid = os.getpid()


class CompositeTSVFile:
    """utils/tsv_file.py:116-158: rows of several TSV files addressed through a sequence file of `source<TAB>row` lines"""

    def __init__(self, file_list, seq_file, root="."):
        self.file_list = load_list_file(file_list) if isinstance(file_list, str) else list(file_list)
        self.seq_file, self.root = seq_file, root
        self.seq = []
        with open(seq_file, "r") as fp:
            for line in fp:
                a, b = line.strip().split("\t")[:2]
                self.seq.append([int(a), int(b)])
        self.tsvs = [TSVFile(op.join(root, f)) for f in self.file_list]

    def get_key(self, index):
        src, row = self.seq[index]
        return "_".join([self.file_list[src], self.tsvs[src].get_key(row)])

    def num_rows(self):
        return len(self.seq)

    __len__ = num_rows

    def __getitem__(self, index):
        src, row = self.seq[index]
        return self.tsvs[src].seek(row)

    def get_composite_source_idx(self):
        return [int(i) for i, _ in self.seq]


# ---- frame columns of a row (dataset.py:136-164) ---------------------------------------------------------------------------------
def sampling(start, end, n):
    """dataset.py:142-146: n frame positions spread over [start, end] (the middle one for n == 1)"""
    if n == 1:
        return [int(round((start + end) / 2.0))]
    if n < 1:
        raise Exception("behaviour not defined for n<2")
    step = (end - start) / float(n - 1)
    return [int(round(start + x * step)) for x in range(n)]


def str2img(b):
    """dataset.py:136-140 (the PIL branch; OpenCV is optional there): base64 image bytes -> PIL RGB image"""
    from PIL import Image
    return Image.open(io.BytesIO(base64.b64decode(b))).convert("RGB")


def frames_to_clip(bufs, size_img, mode="img_center_crop"):
    """The deterministic transforms of dataset.py:91-106 on a list of base64 frames -> (T, 3, size_img, size_img) f32,
    ImageNet-normalised: `img_center_crop` (resize shorter side, centre crop) or `pad_resize` (pad to square, resize)."""
    import numpy as np
    import torch
    from PIL import Image
    mean = np.array([0.485, 0.456, 0.406], np.float32).reshape(3, 1, 1)
    std = np.array([0.229, 0.224, 0.225], np.float32).reshape(3, 1, 1)
    out = []
    for b in bufs:
        im = str2img(b)
        w, h = im.size
        if mode == "pad_resize":
            side = max(w, h)
            canvas = Image.new("RGB", (side, side))
            canvas.paste(im, ((side - w) // 2, (side - h) // 2))
            im = canvas.resize((size_img, size_img), Image.BILINEAR)
        else:
            s = size_img / min(w, h)
            nw, nh = (size_img, int(size_img * h / w)) if w <= h else (int(size_img * w / h), size_img)
            im = im.resize((nw, nh), Image.BILINEAR)
            l, t = int(round((nw - size_img) / 2.0)), int(round((nh - size_img) / 2.0))
            im = im.crop((l, t, l + size_img, t + size_img))
        a = np.asarray(im, dtype=np.float32).transpose(2, 0, 1) / 255.0
        out.append(torch.from_numpy((a - mean) / std))
    return torch.stack(out, 0)
