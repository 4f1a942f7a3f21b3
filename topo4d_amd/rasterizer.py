"""
Host side of the rasterizer: the `diff_gaussian_rasterization` Python surface Topo4D calls, over the C ABI.

Mirrors (names, argument meaning, error behaviour) the un-vendored package the reference imports at
train.py:19 / helpers.py:18-19:

    from diff_gaussian_rasterization import GaussianRasterizer as Renderer
    from diff_gaussian_rasterization import GaussianRasterizationSettings as Camera
    im, radius, depth, alpha = Renderer(raster_settings=cam)(**rendervar)        # train.py:307,388,463,484

plus a multi-view entry point (`rasterize_views`, `ViewBatch`) that renders all cameras of a Topo4D frame —
or one rank's shard of them — in ONE set of kernel launches (grid.z = view): at 30k Gaussians / 512² a single
view is launch-bound on an MI355X, the 24-view batch is not.

PyTorch is plumbing here (device memory, current stream, autograd glue); all arithmetic is in
csrc/t4d_raster.hip behind include/topo4d_raster.h.  There is no CPU path: CPU tensors raise.
"""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict
from typing import List, NamedTuple, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from . import _lib
from ._lib import (T4D_ABI_VERSION, T4D_ERR_PAIR_OVERFLOW, T4D_FLAG_CHECKED, T4D_FLAG_DEBUG_SYNC, T4D_FLAG_NO_LONG_BINS,
                   T4D_FLAG_PREFILTERED, T4D_OK, T4D_VIEW_FLOATS, T4DBackwardIO, T4DForwardIO, T4DProblem,
                   T4DStatus)


class GaussianRasterizationSettings(NamedTuple):
    """Twelve fields, in the order and spelling helpers.py:73-86 constructs them."""
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


# ------------------------------------------------------------------------------------------------------------
# sync policy and pair-arena capacity
# ------------------------------------------------------------------------------------------------------------
_SYNC_MODE = "checked"
_CAPACITY = {}          # (device_index, P, H, W) -> learned per-view pair capacity
_LONGEST_BIN = {}       # (device_index, P, H, W) -> longest tile list a checked forward has reported for this scene size
_AUTO = {}              # (device_index, P, H, W, views key) -> _AutoTrack of the "auto" sync mode
_AUTO_MAX = 256         # camera sets tracked at a time
_BATCH_LOG = None       # when a list: every ViewBatch that runs a forward is appended (loop.GraphedViews keeps the batches
                        # of its captures to read their overflow flags back later)


class _AutoTrack:
    """Bookkeeping of the "auto" sync mode for one (scene size, camera set): a ring of pinned status blocks, one per
    un-synchronised forward still in flight, each with the event that says it has landed, and the largest need seen."""
    RING = 8
    __slots__ = ("pinned", "host", "events", "caps", "head", "count", "need", "key")

    def __init__(self, key=None):
        self.key = key                                  # (device_index, P, H, W) whose capacity this track feeds
        self.pinned = torch.zeros(self.RING, 2, dtype=torch.int64).pin_memory()
        self.host = self.pinned.numpy()                 # same memory, cheap scalar reads
        self.events = [torch.cuda.Event() for _ in range(self.RING)]
        self.caps = [0] * self.RING
        self.head = 0                                   # next slot to hand out
        self.count = 0                                  # slots in flight (oldest = head - count)
        self.need = 0

    def slot_ptr(self) -> int:
        return self.pinned.data_ptr() + 16 * self.head

    def mark(self, stream, cap):
        self.events[self.head].record(stream)
        self.caps[self.head] = cap
        self.head = (self.head + 1) % self.RING
        self.count += 1
        _INFLIGHT[id(self)] = self

    def harvest(self):
        """Yields (overflow, need, capacity used) of every un-synchronised call whose status has landed, oldest first.
        Waits for the oldest one only when the ring is full (the host is then RING forwards ahead of the GPU)."""
        out = []
        while self.count:
            tail = (self.head - self.count) % self.RING
            if self.count == self.RING:
                self.events[tail].synchronize()
            elif not self.events[tail].query():
                break
            raw = int(self.host[tail, 0])
            out.append((raw & 0xffffffff, (raw >> 32) & 0xffffffff, self.caps[tail]))
            self.count -= 1
        if not self.count:
            _INFLIGHT.pop(id(self), None)
        return out


_INFLIGHT = {}          # id(track) -> _AutoTrack with un-harvested status blocks


def poll_truncation() -> None:
    """"auto" sync mode: look at every binning status that has landed since the last look - of ALL camera sets, not only
    the one being rendered - grow the arenas they ask for, and raise RuntimeError if any of those forwards was truncated
    (its backward returned zero gradients, see t4d_rasterize_backward).  Every auto-mode forward calls this first; an
    optimisation loop may also call it right before `optimizer.step()` to learn about a truncated pass as early as the GPU
    allows.  Never synchronises unless a track's ring is full."""
    truncated = None
    for track in list(_INFLIGHT.values()):
        for overflow, need, cap_used in track.harvest():
            track.need = max(track.need, need)
            if overflow:
                truncated = (need, cap_used)
                if track.key is not None:
                    _CAPACITY[track.key] = max(_CAPACITY.get(track.key, 0), _round_capacity(track.need))
    if truncated is not None:
        raise RuntimeError(
            f"topo4d_amd (sync_mode='auto'): an earlier render needed {truncated[0]} (Gaussian,tile) pairs per view but its "
            f"arena held {truncated[1]}; it was truncated and its backward returned zero gradients. The arena has been "
            "enlarged; re-run that iteration, or use set_sync_mode('checked') for scenes that change abruptly.")


def set_sync_mode(mode: str) -> None:
    """"checked" (default): the forward synchronises once, after the binning sizes are known, and re-runs
    with a larger pair arena if it was too small — exactly where upstream reads `num_rendered` back.
    "lazy": never synchronises; tile lists are truncated (memory-safe) if the arena learned by earlier checked
    calls is too small, and `ViewBatch.fetch_status()` / `last_status()` reports it.  Use lazy only when the
    capacity was established by a checked call on (nearly) the same scene — bench.py does.
    "auto": for optimisation loops.  The first forward of every (scene size, camera set) is checked; afterwards the
    forward does not synchronise, the binning status is copied to pinned host memory asynchronously and inspected at the
    next auto-mode forward of ANY camera set (`poll_truncation`; at most 8 calls of one camera set later if the host runs
    ahead of the GPU): the arena is grown as soon as 75 % of it is in use (it is sized 1.5x the largest need seen), so
    consecutive iterations of an optimiser cannot overflow it; should a previous call nevertheless have been truncated
    (the scene jumped by more than a third between two calls), its backward returned zero gradients and a RuntimeError
    says so at the next forward (or at an explicit `poll_truncation()` before `optimizer.step()`)."""
    global _SYNC_MODE
    if mode not in ("checked", "lazy", "auto"):
        raise ValueError("sync mode must be 'checked', 'lazy' or 'auto'")
    _SYNC_MODE = mode


def get_sync_mode() -> str:
    return _SYNC_MODE


def _initial_capacity(P: int) -> int:
    return max(16384, 8 * int(P))


def _round_capacity(n: int) -> int:
    n = int(n * 1.5) + 1024
    return (n + 1023) // 1024 * 1024


# ------------------------------------------------------------------------------------------------------------
# view records
# ------------------------------------------------------------------------------------------------------------
_VIEW_CACHE = OrderedDict()   # id(settings) -> (settings, versions, packed record); least recently used first
_VIEW_CACHE_MAX = 512


def _pack_one_view(s: GaussianRasterizationSettings, device) -> torch.Tensor:
    vm = s.viewmatrix.reshape(-1).to(device=device, dtype=torch.float32)
    pm = s.projmatrix.reshape(-1).to(device=device, dtype=torch.float32)
    if vm.numel() != 16 or pm.numel() != 16:
        raise ValueError("viewmatrix/projmatrix must hold 16 elements ([1,4,4] or [4,4], helpers.py:67,72)")
    cp = s.campos.reshape(-1).to(device=device, dtype=torch.float32)
    bg = s.bg.reshape(-1).to(device=device, dtype=torch.float32)
    if cp.numel() != 3 or bg.numel() != 3:
        raise ValueError("campos and bg must hold 3 elements")
    tan = torch.tensor([float(s.tanfovx), float(s.tanfovy)], dtype=torch.float32).to(device, non_blocking=True)
    return torch.cat([vm, pm, cp, bg, tan])


def pack_views(settings: Sequence[GaussianRasterizationSettings], device) -> torch.Tensor:
    """[V, T4D_VIEW_FLOATS] fp32 device records (layout: include/topo4d_raster.h).  Cached per settings object
    (train.py:98 builds each camera once per frame and reuses it for every iteration)."""
    recs = []
    for s in settings:
        key = id(s)
        ver = (s.viewmatrix._version, s.projmatrix._version, s.campos._version, s.bg._version, str(device))
        hit = _VIEW_CACHE.get(key)
        if hit is not None and hit[0] is s and hit[1] == ver:
            _VIEW_CACHE.move_to_end(key)
            recs.append(hit[2])
            continue
        rec = _pack_one_view(s, device)
        _VIEW_CACHE[key] = (s, ver, rec)
        _VIEW_CACHE.move_to_end(key)
        # evict the least recently used entries only: whoever still holds a record (a ViewBatch, a captured HIP graph via
        # loop.GraphedViews) keeps its own reference, so eviction can never free memory a pending launch reads
        while len(_VIEW_CACHE) > _VIEW_CACHE_MAX:
            _VIEW_CACHE.popitem(last=False)
        recs.append(rec)
    if len(recs) == 1:
        out = recs[0].unsqueeze(0)                           # a view of the cached record: no kernel, no allocation
    else:
        out = torch.stack(recs, 0)
    assert out.shape[1] == T4D_VIEW_FLOATS
    return out


def _check_common(settings: Sequence[GaussianRasterizationSettings]):
    s0 = settings[0]
    for s in settings[1:]:
        if (int(s.image_height), int(s.image_width)) != (int(s0.image_height), int(s0.image_width)):
            raise ValueError("all views of one batch must share image_height/image_width")
        if float(s.scale_modifier) != float(s0.scale_modifier) or int(s.sh_degree) != int(s0.sh_degree):
            raise ValueError("all views of one batch must share scale_modifier and sh_degree")
    return int(s0.image_height), int(s0.image_width), float(s0.scale_modifier), int(s0.sh_degree)


# ------------------------------------------------------------------------------------------------------------
# low-level batch object (no autograd): what bench.py times and the autograd Function drives
# ------------------------------------------------------------------------------------------------------------
def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _f32c(t: Optional[torch.Tensor], name: str, device) -> Optional[torch.Tensor]:
    if t is None or t.numel() == 0:
        return None
    if t.device != device:
        raise ValueError(f"{name} is on {t.device}, expected {device}")
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _raw_stream(device) -> int:
    """hipStream_t of torch's current stream on `device` (the private getter is ~20x cheaper than building a Stream object)."""
    try:
        return torch._C._cuda_getCurrentRawStream(device.index if device.index is not None else torch.cuda.current_device())
    except AttributeError:                                   # pragma: no cover - older/newer torch without the private hook
        return torch.cuda.current_stream(device).cuda_stream


class ViewBatch:
    """One forward (+ optional backward) of V views of the same Gaussians through the C ABI.

    Owns the opaque state buffer between forward and backward, like the (geomBuffer, binningBuffer, imgBuffer)
    byte tensors upstream keeps in its autograd ctx.
    """

    def __init__(self, views: torch.Tensor, H: int, W: int, scale_modifier: float = 1.0, sh_degree: int = 0,
                 debug: bool = False, prefiltered: bool = False, cam_key=None):
        if not views.is_cuda:
            raise RuntimeError("topo4d_amd has no CPU path: tensors must live on a HIP device")
        self.lib = _lib.load()
        self.views = views.contiguous()
        self.V = int(views.shape[0])
        self.H, self.W = int(H), int(W)
        self.scale_modifier = float(scale_modifier)
        self.sh_degree = int(sh_degree)
        self.debug = bool(debug)
        self.prefiltered = bool(prefiltered)
        self.device = views.device
        self.cam_key = cam_key if cam_key is not None else (views.data_ptr(), self.V)   # identity of the camera set ("auto" mode)
        self.state = None
        self.prob = None
        self.inputs = None
        self.radii = None

    # -- helpers ---------------------------------------------------------------------------------------------
    def _problem(self, P: int, M: int, cap: int, checked: bool) -> T4DProblem:
        flags = 0
        if checked:
            flags |= T4D_FLAG_CHECKED
        if self.debug:
            flags |= T4D_FLAG_DEBUG_SYNC
        if self.prefiltered:
            flags |= T4D_FLAG_PREFILTERED
        # tile lists of this scene size stayed well below the LDS sort buffer (2048) so far: skip the long-bin sort launch
        # (a speed hint only - see include/topo4d_raster.h)
        longest = _LONGEST_BIN.get((self.device.index, P, self.H, self.W))
        if longest is not None and longest <= 1536:
            flags |= T4D_FLAG_NO_LONG_BINS
        return T4DProblem(T4D_ABI_VERSION, self.V, P, self.H, self.W, self.sh_degree, M, self.scale_modifier,
                          cap, flags, 0)

    def _stream(self):
        return C.c_void_p(_raw_stream(self.device))

    # -- forward ---------------------------------------------------------------------------------------------
    def forward(self, means3D, opacities, scales=None, rotations=None, colors_precomp=None, shs=None,
                cov3D_precomp=None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
        dev = self.device
        means3D = _f32c(means3D, "means3D", dev)
        if means3D is None:
            raise ValueError("means3D must not be empty")
        P = int(means3D.shape[0])
        opacities = _f32c(opacities, "opacities", dev)
        scales = _f32c(scales, "scales", dev)
        rotations = _f32c(rotations, "rotations", dev)
        colors_precomp = _f32c(colors_precomp, "colors_precomp", dev)
        shs = _f32c(shs, "shs", dev)
        cov3D_precomp = _f32c(cov3D_precomp, "cov3D_precomp", dev)
        if (shs is None) == (colors_precomp is None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        if opacities is None or opacities.numel() != P:
            raise ValueError("opacities must hold one value per Gaussian")
        M = 0
        if shs is not None:
            if shs.dim() != 3 or shs.shape[0] != P or shs.shape[2] != 3:
                raise ValueError("shs must be [P, M, 3]")
            M = int(shs.shape[1])
            if M < (self.sh_degree + 1) ** 2:
                raise ValueError("shs holds fewer coefficients than sh_degree needs")
        V, H, W = self.V, self.H, self.W
        color = torch.empty(V, 3, H, W, dtype=torch.float32, device=dev)
        depth = torch.empty(V, 1, H, W, dtype=torch.float32, device=dev)
        alpha = torch.empty(V, 1, H, W, dtype=torch.float32, device=dev)
        radii = torch.empty(V, P, dtype=torch.int32, device=dev)

        key = (dev.index, P, H, W)
        cap = _CAPACITY.get(key, _initial_capacity(P))
        checked = (_SYNC_MODE == "checked") or self.debug or key not in _CAPACITY
        track = None
        if _SYNC_MODE == "auto" and not self.debug:
            poll_truncation()                                     # statuses of every camera set that have landed by now
            cap = _CAPACITY.get(key, cap)
            akey = key + (self.cam_key,)
            track = _AUTO.get(akey)
            if track is None:
                if len(_AUTO) >= _AUTO_MAX:                       # Topo4D builds new camera tuples every frame (train.py:98):
                    for old in list(_AUTO)[: _AUTO_MAX // 2]:     # forget the oldest half (dicts keep insertion order)
                        del _AUTO[old]
                track = _AUTO[akey] = _AutoTrack(key)
                checked = True                                    # first time these cameras see this scene size
            else:
                if track.need > 0.75 * cap:                       # grow well before the arena can overflow
                    cap = _round_capacity(track.need)
                    _CAPACITY[key] = max(_CAPACITY.get(key, 0), cap)
                cap = _CAPACITY.get(key, cap)
        status = T4DStatus()
        for _attempt in range(6):
            prob = self._problem(P, M, cap, checked)
            status_arg = C.byref(status)
            if track is not None and not checked:
                prob.flags |= _lib.T4D_FLAG_ASYNC_STATUS
                status_arg = C.cast(C.c_void_p(track.slot_ptr()), C.POINTER(T4DStatus))
            nbytes = self.lib.t4d_state_bytes(C.byref(prob))
            state = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            io = T4DForwardIO(_ptr(self.views), _ptr(means3D), _ptr(opacities), _ptr(scales), _ptr(rotations),
                              _ptr(cov3D_precomp), _ptr(colors_precomp), _ptr(shs), _ptr(color), _ptr(depth),
                              _ptr(alpha), _ptr(radii), _ptr(state), nbytes)
            rc = self.lib.t4d_rasterize_forward(C.byref(prob), C.byref(io), status_arg, self._stream())
            if rc == T4D_OK:
                break
            if rc == T4D_ERR_PAIR_OVERFLOW:
                cap = _round_capacity(status.max_pairs_per_view)
                continue
            raise RuntimeError(f"t4d_rasterize_forward failed (code {rc}): {_lib.last_error()}")
        else:
            raise RuntimeError("pair arena kept overflowing; this should be impossible")
        if checked:
            # keep 1.5x head-room over what this scene needs so that lazy calls on nearby scenes fit
            want = _round_capacity(status.max_pairs_per_view)
            _CAPACITY[key] = max(want, cap if key in _CAPACITY else 0)
            _LONGEST_BIN[key] = max(_LONGEST_BIN.get(key, 0), int(status.max_tile_pairs))
            if track is not None:
                track.need = max(track.need, int(status.max_pairs_per_view))
        elif track is not None:
            track.mark(torch.cuda.current_stream(dev), cap)
        self.prob, self.state, self.radii = prob, state, radii
        if _BATCH_LOG is not None:
            _BATCH_LOG.append(self)
        self.inputs = (means3D, opacities, scales, rotations, cov3D_precomp, colors_precomp, shs)
        self.last_status = status if checked else None
        return color, radii, depth, alpha

    # -- backward --------------------------------------------------------------------------------------------
    def backward(self, dL_dcolor, dL_ddepth=None, dL_dalpha=None, cotangent_dot=None):
        """Per-view gradients: dict of [V,P,...] tensors (means3D, means2D, colors_precomp|shs, opacities,
        scales+rotations|cov3D_precomp).  `cotangent_dot`: optional fp32 [V] tensor that receives
        <color, dL_dcolor> + <depth, dL_ddepth> + <alpha, dL_dalpha> per view (a by-product of the replay)."""
        if self.state is None:
            raise RuntimeError("backward() before forward()")
        if self.inputs is None or self.inputs[0] is None:
            raise RuntimeError("this ViewBatch's inputs are owned by an autograd graph (GaussianRasterizer / rasterize_views): "
                               "call .backward() on the loss instead of ViewBatch.backward()")
        dev = self.device
        means3D, opacities, scales, rotations, cov3D_precomp, colors_precomp, shs = self.inputs
        V, P, H, W = self.V, int(means3D.shape[0]), self.H, self.W
        M = 0 if shs is None else int(shs.shape[1])
        dL_dcolor = _f32c(dL_dcolor, "dL_dcolor", dev)
        if dL_dcolor is None or dL_dcolor.numel() != V * 3 * H * W:
            raise ValueError("dL_dcolor must be [V,3,H,W]")
        dL_ddepth = _f32c(dL_ddepth, "dL_ddepth", dev)
        dL_dalpha = _f32c(dL_dalpha, "dL_dalpha", dev)
        if cotangent_dot is not None and (cotangent_dot.dtype != torch.float32 or cotangent_dot.device != dL_dcolor.device
                                          or cotangent_dot.numel() != V or not cotangent_dot.is_contiguous()):
            raise ValueError("cotangent_dot must be a contiguous fp32 [V] tensor on the rasterizer's device")
        # separate allocations on purpose: autograd's AccumulateGrad only adopts a gradient without copying it when the
        # tensor owns its storage (carving them out of one buffer cost six clone kernels per backward)
        new = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)
        g = dict(means3D=new(V, P, 3), means2D=new(V, P, 3), opacities=new(V, P, 1))
        g["colors_precomp"] = new(V, P, 3) if colors_precomp is not None else None
        g["shs"] = new(V, P, M, 3) if shs is not None else None
        g["scales"] = new(V, P, 3) if cov3D_precomp is None else None
        g["rotations"] = new(V, P, 4) if cov3D_precomp is None else None
        g["cov3D_precomp"] = new(V, P, 6) if cov3D_precomp is not None else None
        prob = self.prob
        sbytes = self.lib.t4d_backward_scratch_bytes(C.byref(prob))
        scratch = torch.empty(sbytes, dtype=torch.uint8, device=dev)
        io = T4DBackwardIO(_ptr(self.views), _ptr(means3D), _ptr(opacities), _ptr(scales), _ptr(rotations),
                           _ptr(cov3D_precomp), _ptr(colors_precomp), _ptr(shs), _ptr(self.radii),
                           _ptr(self.state), self.state.numel(), _ptr(dL_dcolor), _ptr(dL_ddepth), _ptr(dL_dalpha),
                           _ptr(g["means3D"]), _ptr(g["means2D"]), _ptr(g["colors_precomp"]), _ptr(g["shs"]),
                           _ptr(g["opacities"]), _ptr(g["scales"]), _ptr(g["rotations"]), _ptr(g["cov3D_precomp"]),
                           _ptr(scratch), sbytes, _ptr(cotangent_dot))
        rc = self.lib.t4d_rasterize_backward(C.byref(prob), C.byref(io), self._stream())
        if rc != T4D_OK:
            raise RuntimeError(f"t4d_rasterize_backward failed (code {rc}): {_lib.last_error()}")
        self._keepalive = (scratch, dL_dcolor, dL_ddepth, dL_dalpha)
        return g

    def fetch_status(self) -> T4DStatus:
        """Synchronising read of the forward's binning status (pairs needed, overflow flag)."""
        st = T4DStatus()
        rc = self.lib.t4d_fetch_status(C.byref(self.prob), _ptr(self.state), C.byref(st), self._stream())
        if rc != T4D_OK:
            raise RuntimeError(f"t4d_fetch_status failed (code {rc}): {_lib.last_error()}")
        return st


def view_dot(a: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[v] = sum(a[v] * b[v]) in one fused, deterministic pass (t4d_view_dot).  a, b: [V, ...] fp32 on the GPU."""
    if not a.is_cuda or a.shape != b.shape or a.dtype != torch.float32 or b.dtype != torch.float32:
        raise ValueError("view_dot needs two fp32 HIP tensors of the same shape")
    lib = _lib.load()
    a, b = a.contiguous(), b.contiguous()
    V = int(a.shape[0])
    n = a.numel() // V
    if out is None:
        out = torch.empty(V, dtype=torch.float32, device=a.device)
    scratch = torch.empty(lib.t4d_view_dot_scratch_bytes(V), dtype=torch.uint8, device=a.device)
    rc = lib.t4d_view_dot(V, n, _ptr(a), _ptr(b), _ptr(out), _ptr(scratch),
                          C.c_void_p(torch.cuda.current_stream(a.device).cuda_stream))
    if rc != T4D_OK:
        raise RuntimeError(f"t4d_view_dot failed (code {rc}): {_lib.last_error()}")
    return out


# ------------------------------------------------------------------------------------------------------------
# autograd glue
# ------------------------------------------------------------------------------------------------------------
class _RasterizeViews(torch.autograd.Function):
    """Inputs in upstream's order; outputs (color[V,3,H,W], radii[V,P], depth[V,1,H,W], alpha[V,1,H,W])."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                views, H, W, scale_modifier, sh_degree, debug, prefiltered, cam_key=None):
        batch = ViewBatch(views, H, W, scale_modifier, sh_degree, debug, prefiltered, cam_key)
        none_if_empty = lambda t: None if (t is None or t.numel() == 0) else t
        color, radii, depth, alpha = batch.forward(
            means3D, opacities, none_if_empty(scales), none_if_empty(rotations), none_if_empty(colors_precomp),
            none_if_empty(sh), none_if_empty(cov3Ds_precomp))
        ctx.batch = batch
        # the backward kernels re-read the forward inputs: saving them through autograd (as upstream does) makes an in-place
        # update between forward and backward raise instead of silently differentiating at the wrong point
        ctx.input_slots = [i for i, t in enumerate(batch.inputs) if t is not None]
        ctx.save_for_backward(*[batch.inputs[i] for i in ctx.input_slots])
        batch.inputs = None                  # ownership moved to the ctx (a direct ViewBatch.backward() now raises clearly)
        ctx.set_materialize_grads(False)     # unused depth/alpha outputs arrive as None -> cheaper backward kernel
        ctx.mark_non_differentiable(radii)
        ctx.shapes = (means3D.shape, means2D.shape if means2D is not None else None, opacities.shape)
        return color, radii, depth, alpha

    @staticmethod
    def backward(ctx, grad_color, grad_radii, grad_depth, grad_alpha):
        batch: ViewBatch = ctx.batch
        inputs = [None] * 7
        for slot, t in zip(ctx.input_slots, ctx.saved_tensors):         # raises if one of them was modified in place
            inputs[slot] = t
        batch.inputs = tuple(inputs)
        if grad_color is None:
            grad_color = torch.zeros(batch.V, 3, batch.H, batch.W, dtype=torch.float32, device=batch.device)
        g = batch.backward(grad_color, grad_depth, grad_alpha)
        batch.inputs = None                  # the batch (kept by GraphedViews / _BATCH_LOG) must not keep the inputs alive
        red = (lambda t: None if t is None else t.sum(0)) if batch.V > 1 else \
              (lambda t: None if t is None else t[0])
        shp3, shp2, shpo = ctx.shapes
        d_means2D = None
        if shp2 is not None and ctx.needs_input_grad[1]:
            d_means2D = red(g["means2D"]).reshape(shp2)
        grads = (
            red(g["means3D"]).reshape(shp3),
            d_means2D,
            red(g["shs"]),
            red(g["colors_precomp"]),
            red(g["opacities"]).reshape(shpo),
            red(g["scales"]),
            red(g["rotations"]),
            red(g["cov3D_precomp"]),
            None, None, None, None, None, None, None, None,
        )
        ctx.batch = None
        return grads


def rasterize_views(settings: Sequence[GaussianRasterizationSettings], means3D, means2D, opacities, shs=None,
                    colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None):
    """Differentiable multi-view render: (color[V,3,H,W], radii[V,P] int32, depth[V,1,H,W], alpha[V,1,H,W]).

    Gradients w.r.t. the shared Gaussian tensors are summed over the V views (the gradient of a loss that
    adds per-view terms), which for V = 1 is the reference's call."""
    if len(settings) == 0:
        raise ValueError("need at least one view")
    if not means3D.is_cuda:
        raise RuntimeError("topo4d_amd has no CPU path: tensors must live on a HIP device")
    H, W, smod, deg = _check_common(settings)
    views = pack_views(settings, means3D.device)
    empty = torch.empty(0, device=means3D.device)
    nz = lambda t: empty if t is None else t
    debug = any(bool(s.debug) for s in settings)
    pref = any(bool(s.prefiltered) for s in settings)
    cam_key = tuple(id(s_) for s_ in settings)           # Settings tuples are built once per camera per frame (train.py:98)
    return _RasterizeViews.apply(means3D, means2D, nz(shs), nz(colors_precomp), opacities, nz(scales),
                                 nz(rotations), nz(cov3D_precomp), views, H, W, smod, deg, debug, pref, cam_key)


class GaussianRasterizer(nn.Module):
    """Drop-in for `diff_gaussian_rasterization.GaussianRasterizer` (constructed per call at train.py:307)."""

    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions: torch.Tensor) -> torch.Tensor:
        """Boolean mask of Gaussians that pass the near-plane test of this camera."""
        with torch.no_grad():
            if not positions.is_cuda:
                raise RuntimeError("topo4d_amd has no CPU path: tensors must live on a HIP device")
            lib = _lib.load()
            pos = positions.float().contiguous()
            views = pack_views([self.raster_settings], pos.device)
            out = torch.empty(pos.shape[0], dtype=torch.uint8, device=pos.device)
            rc = lib.t4d_mark_visible(int(pos.shape[0]), _ptr(pos), _ptr(views), _ptr(out),
                                      C.c_void_p(torch.cuda.current_stream(pos.device).cuda_stream))
            if rc != T4D_OK:
                raise RuntimeError(f"t4d_mark_visible failed (code {rc}): {_lib.last_error()}")
            return out.bool()

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        color, radii, depth, alpha = rasterize_views([self.raster_settings], means3D, means2D, opacities, shs,
                                                     colors_precomp, scales, rotations, cov3D_precomp)
        return color[0], radii[0], depth[0], alpha[0]
