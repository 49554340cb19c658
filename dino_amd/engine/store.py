"""ParamStore: flat fp32 master parameters of one nn.Module tree + the derived device state
the kernels consume (bf16 shadows, W^T shadows for dgrad, the weight-normed prototype matrix).

Every nn.Parameter of the module becomes a VIEW into `flat` (same names / order / shapes, so
state_dict(), load_state_dict() and zip(student.parameters(), teacher.parameters()) behave as
in the reference).  Gradients live in `grad_flat`; p.grad are views of it.
"""
import weakref
from collections import OrderedDict

import torch

from .. import ops
from .flat import FlatLayout


_SIDE_STREAMS = {}


def side_stream(device=None):
    """THE side stream of a device.  The teacher forward (train_utils.train_step) and the grouped
    weight gradients (engine/vit.py) never overlap in time, so they share it: with a stream each
    (three active HIP streams per process) two of five otherwise identical bench runs on the
    MI355X box took 78 ms per step instead of 55.7 (tools/variance.sh, profiles/r01_stream_variance.log)."""
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    if dev is None:
        dev = torch.cuda.current_device()
    if dev not in _SIDE_STREAMS:
        _SIDE_STREAMS[dev] = torch.cuda.Stream(device=dev)
    return _SIDE_STREAMS[dev]


class ParamStore:
    def __init__(self, module: torch.nn.Module, dtype: torch.dtype):
        params = OrderedDict(module.named_parameters())
        if not params:
            raise ValueError("module has no parameters")
        dev = next(iter(params.values())).device
        if dev.type != "cuda":
            from .._lib import DinoHipError
            raise DinoHipError("dino_amd modules run on the GPU only: move the module to cuda "
                               "first (there is no CPU fallback)")
        self.module, self.dtype, self.device = module, dtype, dev
        self.params = params
        self.layout = FlatLayout(OrderedDict((k, tuple(p.shape)) for k, p in params.items()))
        n = self.layout.total
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for k, p in params.items():
                v = self.layout.view(self.flat, k)
                v.copy_(p.data)
                p.data = v
        self.trainable = any(p.requires_grad for p in params.values())
        self.grad_flat = torch.zeros(n, dtype=torch.float32, device=dev) if self.trainable else None
        self.chunks = self.layout.chunks.to(dev)
        self.tensor_chunk_begin = self.layout.tensor_chunk_begin.to(dev)
        self.bf16 = dtype == torch.bfloat16
        self.shadow = torch.zeros(n, dtype=torch.bfloat16, device=dev) if self.bf16 else self.flat
        # W^T shadows of every 2-D weight that needs a dgrad (training only)
        self._tnames = [k for k, p in params.items()
                        if p.dim() == 2 and k.endswith(".weight") and self.trainable]
        t_off, rows = {}, []
        off = 0
        for k in self._tnames:
            r, c = params[k].shape
            t_off[k] = off
            rows.append((self.layout.slices[k][0], off, r, c))
            off += (r * c + 7) // 8 * 8
        self._t_off = t_off
        self.shadow_t = torch.zeros(max(off, 8), dtype=dtype, device=dev)
        self._t_table = torch.tensor(rows, dtype=torch.int64, device=dev).reshape(-1, 4) if rows else None
        # padded patch-embedding weight when 3*P*P is not a multiple of 8 (P = 14)
        self._patch_pad = {}
        # weight-normed prototype layers: name prefix -> (W, W^T, vnorm)
        self._wn = {}
        for k in params:
            if k.endswith("weight_g"):
                self._wn[k[:-len("weight_g")]] = None
        # LayerNorm folded into the GEMM that consumes it (bf16): per (norm, linear) pair of a transformer block the
        # folded weight gamma * W - rowmean(gamma * W) (bf16) and bias b + W beta (fp32), refreshed with the shadows
        self._fold = {}
        self._fold_table = None
        if self.bf16:
            rows, woff, boff, self._fold_maxn = [], 0, 0, 0
            for k in params:
                for norm, lin in (("norm1", "attn.qkv"), ("norm2", "mlp.fc1")):
                    if not k.endswith(f".{norm}.weight") or ".blocks." not in "." + k:
                        continue
                    b = k[:-len(f"{norm}.weight")]
                    wk, bk, btk = f"{b}{lin}.weight", f"{b}{lin}.bias", f"{b}{norm}.bias"
                    if wk not in params or bk not in params or btk not in params or params[wk].dim() != 2:
                        continue
                    N, K = params[wk].shape
                    if K % 4 or params[k].numel() != K:
                        continue
                    sl = self.layout.slices
                    rows.append((sl[wk][0], sl[k][0], sl[btk][0], sl[bk][0], woff, boff, N, K))
                    self._fold[wk] = (woff, boff, N, K)
                    woff += (N * K + 7) // 8 * 8
                    boff += (N + 7) // 8 * 8
                    self._fold_maxn = max(self._fold_maxn, N)
            if rows:
                self._fold_table = torch.tensor(rows, dtype=torch.int64, device=dev).reshape(-1, 8)
                self.fold_w = torch.zeros(woff, dtype=torch.bfloat16, device=dev)
                self.fold_b = torch.zeros(boff, dtype=torch.float32, device=dev)
        self._sig = None
        # tapes of recorded backbone passes that have not been back-propagated yet.  Weak: a forward
        # whose output is dropped without backward (evaluation without no_grad, an exception between
        # forward and backward, a skipped step) frees its tape with the autograd graph and leaves
        # this set by itself instead of blocking every later step's "last pass" detection.
        self.live_tapes = weakref.WeakSet()
        self.ready_cb = None          # DP layer: called with a list of param names whose grad is final
        self.finish_cb = None         # DP layer: called once the whole backward has been issued
        self.anchor = torch.zeros(1, device=dev, requires_grad=True)
        # scratch for the split-K partial tiles of the wgrad kernel (256 CUs x 128x384 fp32 tiles)
        self.workspace = torch.empty(128 << 20, dtype=torch.uint8, device=dev) if self.trainable else None
        self.workspace_wg = torch.empty(128 << 20, dtype=torch.uint8, device=dev) if self.trainable else None
        self.wg_stream = None
        self.wg_pending = []   # operands of weight-gradient launches the main stream has not joined yet
        self.managed = False          # True once FusedAdamW keeps the shadows current itself
        self._need_zero = False       # grad_flat holds a finished step's grads: zero before reuse

    @property
    def pending_bwd(self) -> int:
        """Backbone passes recorded and not yet back-propagated."""
        return len(self.live_tapes)

    # -------------------------------------------------------------------------- freshness
    def _signature(self):
        return (self.flat.data_ptr(), sum(p._version for p in self.params.values()))

    def owns(self) -> bool:
        """True while every parameter still aliases this store's flat buffer."""
        base = self.flat.data_ptr()
        for k, p in self.params.items():
            off, _ = self.layout.slices[k]
            if p.data_ptr() != base + 4 * off or p.device != self.device:
                return False
        return True

    def ensure_fresh(self):
        """Re-derive the shadows if any parameter was modified through PyTorch since the last
        refresh (optimizer.step(), load_state_dict, manual init...).  The fused optimiser kernel
        updates the shadows itself and calls mark_fresh()."""
        sig = self._signature()
        # writes through `p.data` (the reference's EMA loop) do not bump version counters, so an
        # unmanaged frozen module (the teacher) is re-derived on every forward
        if sig != self._sig or (not self.trainable and not self.managed):
            self.refresh_all()

    def mark_fresh(self):
        self._sig = self._signature()

    def refresh_all(self):
        if self.bf16:
            ops.cast_bf16(self.flat, self.shadow)
        self.refresh_derived()

    def refresh_derived(self):
        """Everything computed FROM the (already current) masters/shadows: W^T shadows, the
        weight-normed prototype matrix, padded patch weights."""
        if self._t_table is not None:
            ops.transpose_batched(self.flat, self.shadow_t, self._t_table)
        for pre in self._wn:
            g, v = self.params[pre + "weight_g"], self.params[pre + "weight_v"]
            old = self._wn[pre]
            w, wt, vn = ops.weightnorm_fwd(g.data, v.data, self.dtype,
                                           w=None if old is None else old[0],
                                           wt=None if old is None else old[1],
                                           want_t=self.trainable)
            self._wn[pre] = (w, wt, vn)
        for k, (kp, buf) in self._patch_pad.items():
            w = self.params[k].data
            buf[:, :w[0].numel()].copy_(w.reshape(w.shape[0], -1))
        if self._fold_table is not None:
            ops.ln_fold_weights(self.flat, self._fold_table, self.fold_w, self.fold_b, self._fold_maxn)
        self.mark_fresh()

    # -------------------------------------------------------------------------- accessors
    def w(self, name):
        """Compute-dtype shadow of a parameter, in the parameter's shape."""
        return self.layout.view(self.shadow, name)

    def p32(self, name):
        return self.layout.view(self.flat, name)

    def grad(self, name):
        return self.layout.view(self.grad_flat, name)

    def wt(self, name):
        """[in, out] transposed shadow of a 2-D weight [out, in]."""
        r, c = self.params[name].shape
        off = self._t_off[name]
        return self.shadow_t[off:off + r * c].view(c, r)

    def wfold(self, name):
        """(folded weight bf16 [N, K], folded bias fp32 [N]) of a linear layer behind a LayerNorm, or None."""
        f = self._fold.get(name)
        if f is None:
            return None
        woff, boff, N, K = f
        return self.fold_w[woff:woff + N * K].view(N, K), self.fold_b[boff:boff + N]

    def patch_weight(self, name, kp):
        """[D, kp] view of the conv kernel for the im2col GEMM (zero-padded if 3PP % 8 != 0)."""
        p = self.params[name]
        k = p[0].numel()
        if k == kp:
            return self.w(name).reshape(p.shape[0], k)
        if name not in self._patch_pad:
            buf = torch.zeros((p.shape[0], kp), dtype=self.dtype, device=self.device)
            self._patch_pad[name] = (kp, buf)
            buf[:, :k].copy_(p.data.reshape(p.shape[0], k))
        return self._patch_pad[name][1]

    def wn(self, prefix):
        return self._wn[prefix]

    # -------------------------------------------------------------------------- gradients
    def zero_grad(self):
        if self.grad_flat is not None:
            self.grad_flat.zero_()

    def attach_grads(self, skip=()):
        """Point every trainable parameter's .grad at its slice of grad_flat."""
        for k, p in self.params.items():
            if not p.requires_grad:
                continue
            p.grad = None if k in skip else self.grad(k)

    def begin_recording(self):
        """First recorded forward after a finished backward: the kernels ACCUMULATE into grad_flat,
        so clear it (the optimizer.zero_grad() of train_utils.py:115, whatever set_to_none does)."""
        if self._need_zero:
            self.grad_flat.zero_()
            self._need_zero = False

    def backward_done(self):
        self._need_zero = True
        self.attach_grads()
        if self.finish_cb is not None:
            self.finish_cb()

    def notify_ready(self, names):
        if self.ready_cb is not None:
            self.ready_cb(names)


def store_for(module: torch.nn.Module, dtype: torch.dtype) -> ParamStore:
    """The module's ParamStore, (re)built when the parameters moved (.to(), new dtype)."""
    st = module.__dict__.get("_dino_store")
    if st is None or st.dtype != dtype or not st.owns():
        st = ParamStore(module, dtype)
        module.__dict__["_dino_store"] = st
    return st
