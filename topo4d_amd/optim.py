"""
Fused Adam + region pins — host mirror of the optimiser plumbing Topo4D runs after every rendered view
(reference train.py:272-297 `initialize_optimizer`, helpers.py:801-804 `update_optimizer`, train.py:672-700
`optimizer.step()` followed by the masked "freeze" assignments).

`FusedAdamPins` keeps torch.optim.Adam's surface for what the loop touches (`param_groups` with 'name'/'lr'/'params',
`step()`, `zero_grad(set_to_none)`, `state`) and adds `set_pin(name, index, values)`: rows that must hold fixed values
after every step.  step() is ONE kernel launch (`t4d_adam_pin_step`) for all tensors.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional

import torch

from . import _lib


class FusedAdamPins:
    def __init__(self, param_groups: List[dict], lr: float = 0.0, betas=(0.9, 0.999), eps: float = 1e-15,
                 capturable: bool = False):
        if len(param_groups) > _lib.T4D_ADAM_MAX_TENSORS:
            raise ValueError(f"at most {_lib.T4D_ADAM_MAX_TENSORS} tensors per fused step")
        self.param_groups = []
        for g in param_groups:
            g = dict(g)
            if len(g["params"]) != 1:
                raise ValueError("one tensor per group (as initialize_optimizer builds them, train.py:292-295)")
            g.setdefault("lr", lr)
            g.setdefault("name", f"param{len(self.param_groups)}")
            self.param_groups.append(g)
        self.betas, self.eps = betas, eps
        self.state: Dict[torch.Tensor, dict] = {}
        self._pins: Dict[str, tuple] = {}
        # capturable: step counts and learning rates live in device memory (t4d_adam_pin_step_graph), so that step() can be
        # recorded in a HIP graph and replayed (loop.GraphedViews); call sync_hyper() after changing a group's 'lr'
        self.capturable = capturable
        # names of the tensors whose gradient buffer step() leaves ZEROED (T4D_ADAM_CLEAR_GRAD): persistent buffers that an
        # iteration fills only in part (loop.GraphedViews: the per-camera rows of cam_m / cam_c)
        self.clear_grad: set = set()
        self._step_dev: Optional[torch.Tensor] = None
        self._layout = None                 # (first counter of every tensor, their number) the counter array was built for
        self._lr_dev: Optional[torch.Tensor] = None
        self._lr_host: Optional[List[float]] = None

    def _counter_layout(self):
        """First step counter of every tensor and their total: the fused step keeps one counter per workgroup of 256 elements,
        tensor after tensor in group order (t4d_adam_step_counters)."""
        first, n = [], 0
        for g in self.param_groups:
            first.append(n)
            n += (g["params"][0].numel() + 255) // 256
        return first, n

    def _hyper(self, dev):
        """(step counters, learning rates) on the device.  The counters are laid out by the tensors' CURRENT sizes (one per
        workgroup of 256 elements); when a group's tensor has been replaced by one of another size (the cat_params_to_optimizer /
        remove_points pattern of external.py) the array is rebuilt, every tensor keeping its count - inside a stream capture that is
        impossible (the recorded launches hold the old pointer): then it raises."""
        first, n = self._counter_layout()
        if self._step_dev is None:
            self._step_dev = torch.zeros(max(n, 1), dtype=torch.int32, device=dev)
            self._lr_dev = torch.zeros(len(self.param_groups), dtype=torch.float32, device=dev)
            self._layout = (first, n)
        elif self._layout != (first, n):
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("FusedAdamPins(capturable=True): a parameter changed its size; the recorded steps hold the old "
                                   "counter array - build new graphs")
            old_first, old_n = self._layout
            new = torch.zeros(max(n, 1), dtype=torch.int32, device=dev)
            for k, f in enumerate(first):
                cnt = (first[k + 1] if k + 1 < len(first) else n) - f
                had = (old_first[k + 1] if k + 1 < len(old_first) else old_n) - old_first[k]
                if cnt > 0 and had > 0:
                    new[f:f + cnt] = self._step_dev[old_first[k]]          # all counters of a tensor are equal: any of them
            self._step_dev = new
            self._layout = (first, n)
        return self._step_dev, self._lr_dev

    def sync_hyper(self) -> None:
        """Push the groups' learning rates to the device copy the captured step reads (no-op when nothing changed)."""
        if not self.capturable:
            return
        lrs = [float(g["lr"]) for g in self.param_groups]
        if lrs != self._lr_host:
            _, lr_dev = self._hyper(self.param_groups[0]["params"][0].device)
            lr_dev.copy_(torch.tensor(lrs, dtype=torch.float32), non_blocking=False)
            self._lr_host = lrs

    def steps(self) -> List[int]:
        """Per-tensor step counts (synchronising read in capturable mode)."""
        if self.capturable and self._step_dev is not None:
            self._hyper(self._step_dev.device)              # (re-laid-out first if a tensor changed its size)
            first, _ = self._layout
            host = self._step_dev.tolist()
            return [int(host[f]) if g["params"][0].numel() > 0 else 0 for f, g in zip(first, self.param_groups)]
        return [int(self.state.get(g["params"][0], {}).get("step", 0)) for g in self.param_groups]

    # -- pins ------------------------------------------------------------------------------------------------
    def set_pin(self, name: str, index, values) -> None:
        """After every step, rows `index` (bool mask or integer indices) of parameter `name` hold `values`
        (broadcastable to [n_selected, width]).  Later calls for the same name override earlier ones on the rows they
        share — the order of the assignments in train.py:676-700."""
        p = self._param(name)
        rows, width = p.shape[0], p[0].numel() if p.dim() > 1 else 1
        mask, vals = self._pins.get(name, (None, None))
        if mask is None:
            mask = torch.zeros(rows, dtype=torch.uint8, device=p.device)
            vals = torch.zeros(rows, width, dtype=torch.float32, device=p.device)
        idx = torch.as_tensor(index, device=p.device)
        if idx.dtype == torch.bool:
            sel = idx
        else:
            sel = torch.zeros(rows, dtype=torch.bool, device=p.device)
            sel[idx.long()] = True
        n_sel = int(sel.sum())
        vt = torch.as_tensor(values, dtype=torch.float32, device=p.device)
        if vt.numel() == n_sel * width:
            vt = vt.reshape(n_sel, width)                       # one row of values per selected row, in index order
            if idx.dtype != torch.bool:                         # integer indices may be unsorted: place by index
                full = torch.zeros(rows, width, dtype=torch.float32, device=p.device)
                full[idx.long()] = vt
                vt = full[sel]
        else:
            vt = torch.broadcast_to(vt.reshape(-1, width) if vt.numel() >= width else vt.reshape(-1, 1), (n_sel, width))
        mask = mask.clone()
        vals = vals.clone()
        mask[sel] = 1
        vals[sel] = vt
        self._pins[name] = (mask, vals)

    def clear_pin(self, name: Optional[str] = None) -> None:
        if name is None:
            self._pins.clear()
        else:
            self._pins.pop(name, None)

    def _param(self, name: str) -> torch.Tensor:
        for g in self.param_groups:
            if g["name"] == name:
                return g["params"][0]
        raise KeyError(name)

    # -- torch.optim surface ------------------------------------------------------------------------------------
    def zero_grad(self, set_to_none: bool = True) -> None:
        for g in self.param_groups:
            p = g["params"][0]
            if p.grad is not None:
                if set_to_none:
                    p.grad = None
                else:
                    p.grad.zero_()

    @torch.no_grad()
    def apply_pins(self, names=None) -> None:
        """Write the pinned rows of the named tensors (default: every tensor that has pins) WITHOUT an optimiser step: ONE launch
        of the same kernel with no gradients.  The texture loop pins dense_rgb_colors BEFORE each render and not after the step
        (train.py:731-734), the geometry loop after it (train.py:676-700: step())."""
        lib = _lib.load()
        todo = [g for g in self.param_groups if g["name"] in self._pins and (names is None or g["name"] in names)]
        if not todo:
            return
        arr = (_lib.T4DAdamTensor * len(todo))()
        ptr = lambda t: C.c_void_p(t.data_ptr())
        for k, g in enumerate(todo):
            p = g["params"][0]
            if not p.is_cuda:
                raise RuntimeError("topo4d_amd has no CPU path: parameters must live on a HIP device")
            if not p.is_contiguous() or p.dtype != torch.float32:
                raise ValueError("parameters must be contiguous float32")
            rows = p.shape[0] if p.dim() > 0 else 1
            mask, vals = self._pins[g["name"]]
            arr[k] = _lib.T4DAdamTensor(ptr(p), None, None, None, ptr(mask), ptr(vals), rows, p.numel() // max(rows, 1), 0.0, 0, 0)
        dev = todo[0]["params"][0].device
        rc = lib.t4d_adam_pin_step(arr, len(todo), float(self.betas[0]), float(self.betas[1]), float(self.eps),
                                   C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        if rc != 0:
            raise RuntimeError(f"t4d_adam_pin_step failed (code {rc}): {_lib.last_error()}")

    @torch.no_grad()
    def step(self, pins: bool = True) -> None:
        """Adam for every tensor that has a gradient, then (pins=True) the pinned rows of every tensor - one launch."""
        lib = _lib.load()
        from . import rasterizer
        if rasterizer._PENDING and not torch.cuda.is_current_stream_capturing():
            # "auto" sync mode of the rasterizer: a render whose pair arena overflowed raises HERE at the latest, before the step
            # (its backward carried zero gradients); nothing pending: one truth test of a deque
            rasterizer.poll_truncation()
        arr = (_lib.T4DAdamTensor * len(self.param_groups))()
        keep = []
        dev = None
        for k, g in enumerate(self.param_groups):
            p = g["params"][0]
            if not p.is_cuda:
                raise RuntimeError("topo4d_amd has no CPU path: parameters must live on a HIP device")
            if not p.is_contiguous() or p.dtype != torch.float32:
                raise ValueError("parameters must be contiguous float32")
            dev = p.device
            rows = p.shape[0] if p.dim() > 0 else 1
            width = p.numel() // max(rows, 1)
            grad = None
            if p.grad is not None:
                grad = p.grad.contiguous()
                st = self.state.setdefault(p, {})
                if "exp_avg" not in st:
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                st["step"] = st.get("step", 0) + 1           # per parameter, as torch.optim.Adam counts it
                keep.append(grad)
            st = self.state.get(p, {})
            mask, vals = self._pins.get(g["name"], (None, None)) if pins else (None, None)
            ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr())
            arr[k] = _lib.T4DAdamTensor(ptr(p), ptr(grad), ptr(st.get("exp_avg")) if grad is not None else None,
                                        ptr(st.get("exp_avg_sq")) if grad is not None else None, ptr(mask), ptr(vals), rows,
                                        width, float(g["lr"]), int(st.get("step", 0)) if grad is not None else 0,
                                        _lib.T4D_ADAM_CLEAR_GRAD if (grad is not None and g["name"] in self.clear_grad) else 0)
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        if self.capturable:
            if not torch.cuda.is_current_stream_capturing():
                self.sync_hyper()
            step_dev, lr_dev = self._hyper(dev)
            rc = lib.t4d_adam_pin_step_graph(arr, len(self.param_groups), float(self.betas[0]), float(self.betas[1]),
                                             float(self.eps), C.c_void_p(step_dev.data_ptr()), self._layout[1], C.c_void_p(lr_dev.data_ptr()), stream)
        else:
            rc = lib.t4d_adam_pin_step(arr, len(self.param_groups), float(self.betas[0]), float(self.betas[1]),
                                       float(self.eps), stream)
        if rc != 0:
            raise RuntimeError(f"t4d_adam_pin_step failed (code {rc}): {_lib.last_error()}")
