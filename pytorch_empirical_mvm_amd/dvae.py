"""Frozen DALL-E dVAE tokenizer for the MVM 'vq' target.

Reference: `DalleModel` visbackbone/dalle/__init__.py:23-58, `Encoder` encoder.py:41-93, `EncoderBlock` :12-39,
`Conv2d` utils.py:10-43, `map_pixels` :46-52.  The teacher runs without gradient and only produces integer targets.

Two execution paths:
* native (default on the GPU when every channel count is a multiple of 64, i.e. the real n_hid = 256 encoder): all 3x3
  convolutions run as IMPLICIT GEMMs on the persistent MFMA kernel of libvmvm (`vmvm_gemm_desc.conv_taps = 9`, fp16 operands,
  NHWC activations, zero padding through out-of-range DMA offsets) and all 1x1 convolutions as plain fp16 GEMMs with the
  bias / ReLU / post_gain-scaled residual epilogues; the 7x7 stem is an im2col kernel (pixel pre-processing fused) + GEMM, the
  block-input ReLU is applied to the A fragments of conv_1 (`a_relu`), the max-pools are an NHWC kernel and the arg-max is fused
  into the output convolution's epilogue (the 8192-wide logits are never written) -- no PyTorch compute op is left in this path.
* torch (`F.conv2d`, CPU or reduced test encoders): the first pass SURVEY 8f.1 prescribes, kept as the reference path."""
import math

import torch
import torch.nn.functional as F

IMNET_MEAN = (0.485, 0.456, 0.406)
IMNET_STD = (0.229, 0.224, 0.225)


def param_shapes(n_hid=256, vocab=8192):
    """state_dict keys of the reference Encoder (prefix `dalle.encoder.` inside VIOLET_Pretrain)."""
    s = {}
    pre = "blocks."
    s[pre + "input.w"] = (n_hid, 3, 7, 7); s[pre + "input.b"] = (n_hid,)
    n_in = n_hid
    for gi, mult in enumerate((1, 2, 4, 8)):
        n_out = mult * n_hid
        for bi in range(2):
            q = pre + f"group_{gi + 1}.block_{bi + 1}."
            if n_in != n_out:
                s[q + "id_path.w"] = (n_out, n_in, 1, 1); s[q + "id_path.b"] = (n_out,)
            hid = n_out // 4
            s[q + "res_path.conv_1.w"] = (hid, n_in, 3, 3); s[q + "res_path.conv_1.b"] = (hid,)
            s[q + "res_path.conv_2.w"] = (hid, hid, 3, 3); s[q + "res_path.conv_2.b"] = (hid,)
            s[q + "res_path.conv_3.w"] = (hid, hid, 3, 3); s[q + "res_path.conv_3.b"] = (hid,)
            s[q + "res_path.conv_4.w"] = (n_out, hid, 1, 1); s[q + "res_path.conv_4.b"] = (n_out,)
            n_in = n_out
    s[pre + "output.conv.w"] = (vocab, 8 * n_hid, 1, 1); s[pre + "output.conv.b"] = (vocab,)
    return s


class DalleTeacher:
    """Holds the frozen encoder weights (f32 masters + compute-dtype copies) and extracts token maps."""

    def __init__(self, n_hid=256, vocab=8192, device="cuda", dtype=None, seed=0):
        self.n_hid, self.vocab, self.device = n_hid, vocab, torch.device(device)
        self.dtype = dtype or (torch.float16 if self.device.type == "cuda" else torch.float32)
        g = torch.Generator(device="cpu").manual_seed(seed)
        self.w = {}
        for k, shp in param_shapes(n_hid, vocab).items():
            if k.endswith(".w"):                                         # utils.py:28 : normal(std = 1/sqrt(n_in * kw^2)) ; biases zero
                t = torch.randn(shp, generator=g) / math.sqrt(shp[1] * shp[2] * shp[3])
            else:
                t = torch.zeros(shp)
            self.w[k] = t.to(self.device)
        self.native = self.device.type == "cuda" and n_hid % 256 == 0 and vocab % 64 == 0 and self.dtype == torch.float16
        self._nw = None
        self.channels_last = self.device.type == "cuda"      # NHWC convolutions: 252 -> 194 ms for 256 frames on MI355X (tools/bench_teacher.py)
        self._refresh()

    def _native_weights(self):
        """fp16 GEMM operands: 3x3 kernels as [C_out][9*C_in] with k = tap*C_in + c (tap = ky*3 + kx), 1x1 as [C_out][C_in], the 7x7
        stem as [C_out][192] with k = ky*24 + kx*3 + c (the column order of vmvm_dvae_stem_im2col, zero columns where it pads)"""
        if self._nw is None:
            nw = {}
            for k, v in self.w.items():
                if not k.endswith(".w"):
                    continue
                if k.startswith("blocks.input"):
                    w = torch.zeros((v.shape[0], 7, 8, 3), device=v.device, dtype=torch.float32)
                    w[:, :, :7, :] = v.permute(0, 2, 3, 1)                                     # [co][ky][kx][c]
                    nw[k] = torch.nn.functional.pad(w.reshape(v.shape[0], 168), (0, 24)).to(torch.float16).contiguous()
                else:
                    nw[k] = v.permute(0, 2, 3, 1).reshape(v.shape[0], -1).to(torch.float16).contiguous()
            self._nw = nw
        return self._nw

    @torch.no_grad()
    def _trunk_native(self, img):
        """img (N,3,H,W) f32 ImageNet-normalised -> the last block's output x, fp16 NHWC rows [N*(H/8)*(W/8), 8*n_hid] (before the
        output ReLU).  Every pass is a libvmvm launch: pre-processing + stem im2col, GEMMs (3x3 convolutions implicit; the block-input
        ReLU is applied to conv_1's A fragments, `a_relu`), 2x2 max-pools."""
        from . import kernels as K
        nw = self._native_weights()
        post_gain = 1.0 / (4 * 2) ** 2
        N, _, H, W = img.shape
        x = K.gemm(K.dvae_stem_im2col(img.contiguous()), nw["blocks.input.w"], bias=self.w["blocks.input.b"], fp16=True)   # NHWC rows
        for gi in range(4):
            for bi in range(2):
                q = f"blocks.group_{gi + 1}.block_{bi + 1}."
                idp = K.gemm(x, nw[q + "id_path.w"], bias=self.w[q + "id_path.b"], fp16=True) if (q + "id_path.w") in nw else x
                r = K.gemm(x, nw[q + "res_path.conv_1.w"], bias=self.w[q + "res_path.conv_1.b"], act=2, fp16=True, conv=(9, H, W), a_relu=True)
                r = K.gemm(r, nw[q + "res_path.conv_2.w"], bias=self.w[q + "res_path.conv_2.b"], act=2, fp16=True, conv=(9, H, W))
                r = K.gemm(r, nw[q + "res_path.conv_3.w"], bias=self.w[q + "res_path.conv_3.b"], act=2, fp16=True, conv=(9, H, W))
                Co = nw[q + "res_path.conv_4.w"].shape[0]
                x = K.gemm(r, nw[q + "res_path.conv_4.w"], bias=self.w[q + "res_path.conv_4.b"], col_scale=post_gain, col_scale_n=Co,
                           resid=idp, fp16=True)                                               # id + post_gain * (conv_4 + b)
            if gi < 3:
                x = K.maxpool2x2_nhwc(x, N, H, W)
                H, W = H // 2, W // 2
        return x

    @torch.no_grad()
    def logits_native(self, img):
        """img (N,3,H,W) f32 ImageNet-normalised -> logits f32 [N*(H/8)*(W/8), vocab] (NHWC row order).  Test surface: the
        training path never writes the logits (tokens_native)."""
        from . import kernels as K
        x = self._trunk_native(img)
        return K.gemm(x, self._native_weights()["blocks.output.conv.w"], bias=self.w["blocks.output.conv.b"], out_dtype=torch.float32, fp16=True,
                      a_relu=True)

    @torch.no_grad()
    def tokens_native(self, img):
        """img (N,3,H,W) f32 ImageNet-normalised -> token ids int64 [N*(H/8)*(W/8)]: the output convolution's epilogue keeps one
        (maximum, column) pair per 64 logits, vmvm_argmax_pairs picks the winner -- torch.argmax(z_logits, 1) without the logits"""
        from . import kernels as K
        x = self._trunk_native(img)
        groups = self.vocab // 64
        pairs = torch.empty((x.shape[0], 2 * groups), device=x.device, dtype=torch.float32)
        K.gemm(x, self._native_weights()["blocks.output.conv.w"], bias=self.w["blocks.output.conv.b"], out=pairs, N=self.vocab, act=5, fp16=True, a_relu=True)
        return K.argmax_pairs(pairs, groups)

    def _refresh(self):
        self._nw = None
        self.c = {k: (v if k.startswith("blocks.output") else v.to(self.dtype)) for k, v in self.w.items()}    # last conv stays f32 (encoder.py:72)

    def state_dict(self, prefix="dalle.encoder."):
        return {prefix + k: v.detach().cpu() for k, v in self.w.items()}

    def load_state_dict(self, sd, prefix="dalle.encoder."):
        for k in self.w:
            if prefix + k in sd:
                self.w[k] = sd[prefix + k].to(self.device, torch.float32).reshape(self.w[k].shape)
        self._refresh()

    def load_pickle(self, path):
        """`dalle_model_path` (visbackbone/dalle/__init__.py:12-20): the reference torch.load()s a pickled dall_e Encoder module.
        Accepts that module (anything with .state_dict()) or a plain state_dict saved from it; keys `blocks.*`."""
        obj = torch.load(path, map_location="cpu", weights_only=False)
        sd = obj.state_dict() if hasattr(obj, "state_dict") else obj
        self.load_state_dict({("dalle.encoder." + k): v for k, v in sd.items()})

    def _conv(self, x, k):
        w = self.c[k + ".w"]
        return F.conv2d(x.to(w.dtype), w, self.c[k + ".b"], padding=(w.shape[-1] - 1) // 2)

    @torch.no_grad()
    def logits(self, x):
        """x: (N,3,H,W) f32 already un-normalised and pixel-mapped -> (N, vocab, H/8, W/8) f32"""
        post_gain = 1.0 / (4 * 2) ** 2
        if self.channels_last:
            x = x.contiguous(memory_format=torch.channels_last)
        x = self._conv(x, "blocks.input")
        for gi in range(4):
            for bi in range(2):
                q = f"blocks.group_{gi + 1}.block_{bi + 1}."
                idp = self._conv(x, q + "id_path") if (q + "id_path.w") in self.c else x
                r = self._conv(F.relu(x), q + "res_path.conv_1")
                r = self._conv(F.relu(r), q + "res_path.conv_2")
                r = self._conv(F.relu(r), q + "res_path.conv_3")
                r = self._conv(F.relu(r), q + "res_path.conv_4")
                x = idp + post_gain * r
            if gi < 3:
                x = F.max_pool2d(x, 2)
        return self._conv(F.relu(x).float(), "blocks.output.conv")

    @torch.no_grad()
    def extract_vq_token(self, img, chunk=32):
        """DalleModel.extract_vq_token (__init__.py:44-54): img (N,3,H,W) ImageNet-normalised f32 -> (N, H/8, W/8) int64"""
        mean = torch.tensor(IMNET_MEAN, device=img.device).view(1, 3, 1, 1)
        std = torch.tensor(IMNET_STD, device=img.device).view(1, 3, 1, 1)
        out = []
        if self.native:
            chunk = min(chunk * 2, 64)                                   # 32-bit DMA offsets: <= 2 GiB per activation tensor
        for a in range(0, img.shape[0], chunk):                          # bounded activation memory (224^2 x 256 ch per frame)
            if self.native:
                xi = img[a:a + chunk].float()
                n, hv, wv = xi.shape[0], xi.shape[2] // 8, xi.shape[3] // 8
                out.append(self.tokens_native(xi).view(n, hv, wv))
            else:
                x = img[a:a + chunk].float() * std + mean
                x = 0.8 * x + 0.1                                        # map_pixels, logit_laplace_eps = 0.1
                out.append(torch.argmax(self.logits(x), dim=1))
        return torch.cat(out, 0)

    def get_vq_patch_size(self):
        return 8
