"""TSV container of the reference's data pipeline (`utils/tsv_file.py:43-159`, `dataset.py:36-38,136-164`): one sample per
line, tab-separated columns (key, ..., base64-encoded frames), random access through a `.lineidx` side file of byte offsets; a
composite view over several files through a (source, row) sequence file.  Host-side format code, no GPU involved -- it sits on the
input side of the step (`img`, `txt`, `mask` batches) and keeps the reference's call surface (`TSVFile(path)[i]`, `num_rows`,
`seek`, `seek_first_column`, `get_key`, `CompositeTSVFile`, `create_lineidx`, `load_list_file`)."""
import base64
import io
import mmap
import os

import numpy as np


# The format (what `utils/tsv_file.py` reads and writes): a TSV file is UTF-8 text, one sample per '\n'-terminated line, columns
# separated by '\t'; `<stem>.lineidx` is text, line i = the decimal byte offset at which sample i starts.  This reader maps the TSV
# file into memory once per process and keeps the offsets as one int64 array: sample i is the byte range [off[i], off[i+1]) of the
# map -- no seek / readline state, nothing to re-open after a DataLoader fork (a map is inherited copy-free), O(1) per access.
def _line_starts(buf, size):
    """byte offset of every line start of a mapped file (the last line may lack its newline)"""
    nl = np.flatnonzero(np.frombuffer(buf, dtype=np.uint8, count=size) == 10).astype(np.int64) + 1
    starts = np.concatenate([np.zeros(1, np.int64), nl])
    return starts[starts < size]


def create_lineidx(filein, idxout):
    """Write the `.lineidx` side file of `filein` (same content as utils/tsv_file.py:18-27 produces); the file appears atomically."""
    size = os.path.getsize(filein)
    if size == 0:
        starts = np.zeros(0, np.int64)
    else:
        with open(filein, "rb") as f, mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ) as mm:
            starts = _line_starts(mm, size)
    tmp = f"{idxout}.{os.getpid()}.part"
    with open(tmp, "w") as f:
        f.write("".join(f"{int(o)}\n" for o in starts))
    os.replace(tmp, idxout)


def load_list_file(fname):
    """utils/tsv_file.py:151-157: one entry per line, surrounding blanks dropped, a trailing empty line ignored"""
    with open(fname, "r") as f:
        entries = [ln.strip() for ln in f.read().split("\n")]
    while entries and entries[-1] == "":
        entries.pop()
    return entries


class TSVFile:
    """Random access to the rows of one TSV file (call surface of utils/tsv_file.py:43-113)."""

    def __init__(self, tsv_file, generate_lineidx=False):
        self.tsv_file = tsv_file
        self.lineidx = os.path.splitext(tsv_file)[0] + ".lineidx"
        self._map = None             # (mmap, size), opened on first access
        self._off = None             # int64 [num_rows + 1]: row starts + end of file
        if generate_lineidx and not os.path.isfile(self.lineidx):
            create_lineidx(self.tsv_file, self.lineidx)

    def __repr__(self):
        return f"TSVFile(tsv_file='{self.tsv_file}')"

    __str__ = __repr__

    def _offsets(self):
        if self._off is None:
            with open(self.lineidx, "r") as f:
                off = np.array(f.read().split(), dtype=np.int64)
            self._off = np.concatenate([off, np.array([os.path.getsize(self.tsv_file)], np.int64)])
        return self._off

    def _bytes(self, idx, stop_at=None):
        off = self._offsets()
        n = len(off) - 1
        if not -n <= idx < n:
            raise IndexError(idx)
        idx %= n
        if self._map is None:
            with open(self.tsv_file, "rb") as f:
                self._map = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
        a, b = int(off[idx]), int(off[idx + 1])
        if stop_at is not None:
            e = self._map.find(stop_at, a, b)
            b = e if e >= 0 else b
        return self._map[a:b]

    def num_rows(self):
        return len(self._offsets()) - 1

    __len__ = num_rows

    def seek(self, idx):
        """row `idx` as its list of columns, each stripped of surrounding blanks"""
        line = self._bytes(idx, b"\n").decode("utf-8")
        return [c.strip() for c in line.split("\t")]

    __getitem__ = seek

    def seek_first_column(self, idx):
        """the key (first column) of row `idx` without decoding the payload columns"""
        return self._bytes(idx, b"\t").decode("utf-8").rstrip("\n")      # (a row without a tab is its own key)

    get_key = seek_first_column


class CompositeTSVFile:
    """Rows of several TSV files addressed through a sequence file of `source<TAB>row` lines (utils/tsv_file.py:116-158)."""

    def __init__(self, file_list, seq_file, root="."):
        self.file_list = load_list_file(file_list) if isinstance(file_list, str) else list(file_list)
        self.seq_file, self.root = seq_file, root
        seq = np.loadtxt(seq_file, dtype=np.int64, delimiter="\t", usecols=(0, 1), ndmin=2)
        self.seq = seq.tolist()
        self.tsvs = [TSVFile(os.path.join(root, f)) for f in self.file_list]

    def num_rows(self):
        return len(self.seq)

    __len__ = num_rows

    def __getitem__(self, index):
        src, row = self.seq[index]
        return self.tsvs[src].seek(row)

    def get_key(self, index):
        src, row = self.seq[index]
        return f"{self.file_list[src]}_{self.tsvs[src].get_key(row)}"

    def get_composite_source_idx(self):
        return [src for src, _ in self.seq]


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
