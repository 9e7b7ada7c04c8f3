"""Host-side mirror of the reference's model interface for the inference path.

``DanNet`` stands where ``Basic2DNet`` stands in the reference (dl4vc/model.py:31-961): it is built
from the same structural flags, takes a reference-format ``state_dict`` (``module.``-prefixed keys
accepted, main.py:117,196), and is called with the same arguments as ``trainer.test`` calls the model
(dl4vc/trainer.py:569-572).  All arithmetic happens in the HIP library (``dl4vc_amd.capi``); there is
no PyTorch/CPU fallback and construction fails loudly without the extension or a GPU.
"""
from __future__ import annotations

from typing import Dict, Mapping, Optional

import numpy as np

from .capi import DanHandle
from .config import DanConfig

_IGNORED_KEYS = ("num_batches_tracked",)
# learnable scalars the reference returns from forward (model.py:429-431) but never uses in inference
_PASSTHROUGH = ("bin_output_weights", "vt_output_weights")


def normalise_state_dict(state_dict: Mapping[str, object]) -> Dict[str, np.ndarray]:
    """Reference checkpoint keys -> C-ABI tensor names (fp32 numpy).

    * strips the DataParallel ``module.`` prefix (main.py:117,196);
    * drops BatchNorm's ``num_batches_tracked`` counters;
    * renames the FC Linear layers by STRUCTURE: their ``conv2hidden.<i>`` index is 1/4 when the model was
      built with dropout and 0/3 without (model.py:369-377)."""
    out: Dict[str, np.ndarray] = {}
    fc_idx = set()
    for k, v in state_dict.items():
        if k.startswith("module."):
            k = k[len("module."):]
        if k.endswith(_IGNORED_KEYS):
            continue
        a = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
        if k.startswith("conv2hidden."):
            fc_idx.add(int(k.split(".")[1]))
        out[k] = np.ascontiguousarray(a, dtype=np.float32)
    for new, old in enumerate(sorted(fc_idx)):
        for part in ("weight", "bias"):
            out["fc.%d.%s" % (new, part)] = out.pop("conv2hidden.%d.%s" % (old, part))
    return out


def load_checkpoint(path: str) -> Dict[str, np.ndarray]:
    """``torch.load(path, map_location='cpu')['state_dict']`` (main.py:121-124) -- the only torch use."""
    import torch
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    sd = ckpt["state_dict"] if isinstance(ckpt, dict) and "state_dict" in ckpt else ckpt
    return normalise_state_dict(sd)


class DanNet:
    def __init__(self, config: DanConfig, device_id: int = 0, max_batch: int = 0, chunk_sites: int = 0):
        self.config = config
        self.handle = DanHandle(config, device_id=device_id, max_batch=max_batch, chunk_sites=chunk_sites)
        self._loaded = False
        self._scalars: Dict[str, np.ndarray] = {}

    # ---- lifecycle (Basic2DNet(...) -> load_state_dict -> eval) ---------------------------------
    def load_state_dict(self, state_dict: Mapping[str, object]) -> "DanNet":
        if self._loaded:
            raise RuntimeError("weights already loaded; create a new DanNet")
        sd = normalise_state_dict(state_dict)
        for k in _PASSTHROUGH:
            if k in sd:
                self._scalars[k] = sd.pop(k)
        for k, v in sd.items():
            self.handle.set_tensor(k, v)
        self.handle.finalize()            # validates every shape against the config, fails loudly
        self._loaded = True
        return self

    def eval(self) -> "DanNet":           # inference only: BN uses running stats, dropout is identity
        return self

    def close(self):
        self.handle.close()

    # ---- native call: uint8 planes in HDF5 order ------------------------------------------------
    def forward_u8(self, reads, qual, strand, ref, ref_mask, var_mask, aux: bool = False) -> Dict[str, np.ndarray]:
        """reads/qual/strand ``[B][R][L]``, ref/ref_mask/var_mask ``[B][L]`` uint8 ->
        ``bin_logits (B,2) vt_logits (B,3) vt_prob (B,3)=(NV,HV,OV) bp (B,)`` (+ af/cov/vb/vr with ``aux``)."""
        if not self._loaded:
            raise RuntimeError("load_state_dict() first")
        return self.handle.forward(reads, qual, strand, ref, ref_mask, var_mask, aux=aux)

    def forward_u8_async(self, reads, qual, strand, ref, ref_mask, var_mask, aux: bool = False):
        """Enqueue a batch (<= max_batch sites); returns a token for ``wait``.  Two batches may be in flight: H2D of batch
        k+1 and the caller's post-processing of batch k-1 overlap the forward of batch k (include/dl4vc_dan.h)."""
        if not self._loaded:
            raise RuntimeError("load_state_dict() first")
        return self.handle.forward_async(reads, qual, strand, ref, ref_mask, var_mask, aux=aux)

    def wait(self, token) -> Dict[str, np.ndarray]:
        return self.handle.wait(token)

    # ---- reference-compatible call (trainer.py:569-572) -------------------------------------------
    def __call__(self, reads, ref, q_scores=None, strands=None, binary_trust_vector=None, af_scores=None,
                 ref_bases=None, var_bases=None, ref_masks=None, var_masks=None,
                 rm_non_var_reads=0, rm_var_reads=0, debug=False):
        """Same signature and 14-tuple as ``Basic2DNet.forward`` (model.py:434-436, 959-961).  ``reads``,
        ``q_scores``, ``strands`` arrive as the reference builds them, ``(B, L, R)`` integer tensors;
        ``binary_trust_vector``, ``af_scores``, ``ref_bases``, ``var_bases`` are accepted and unused exactly
        as in the reference's supported configuration."""
        if rm_non_var_reads or rm_var_reads:
            raise NotImplementedError("training-time read removal (model.py:633-716) is out of scope")

        def plane(x, name):
            if x is None:
                raise ValueError("%s is required" % name)
            a = x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)
            if a.min(initial=0) < 0 or a.max(initial=0) > 255:
                raise ValueError("%s out of uint8 range" % name)
            return a.astype(np.uint8)

        rd = np.ascontiguousarray(np.transpose(plane(reads, "reads"), (0, 2, 1)))
        B, R, L = rd.shape
        ql = (np.ascontiguousarray(np.transpose(plane(q_scores, "q_scores"), (0, 2, 1)))
              if self.config.use_q else np.zeros_like(rd))
        st = (np.ascontiguousarray(np.transpose(plane(strands, "strands"), (0, 2, 1)))
              if self.config.use_strand else np.zeros_like(rd))
        zeros = np.zeros((B, L), np.uint8)
        rm = plane(ref_masks, "ref_masks") if self.config.use_mask else zeros
        vm = plane(var_masks, "var_masks") if self.config.use_mask else zeros
        o = self.forward_u8(rd, ql, st, plane(ref, "ref"), rm, vm, aux=True)
        bw = self._scalars.get("bin_output_weights", np.full((1,), 0.1, np.float32))
        vw = self._scalars.get("vt_output_weights", np.full((1,), 0.1, np.float32))
        return (o["bin_logits"], o["vt_logits"], o["af"], o["cov"], o["vb"], o["vr"], [], [], bw, vw,
                None, None, None, None)

    forward = __call__
