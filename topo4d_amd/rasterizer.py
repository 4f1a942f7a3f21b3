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
from collections import OrderedDict, deque
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
_SYNC_MODE_EXPLICIT = False     # set_sync_mode() was called: until then the one-view drop-in (GaussianRasterizer) runs "auto"
_BATCH_LOG = None       # when a list: every ViewBatch that runs a forward is appended (loop.GraphedViews keeps the batches
                        # of its captures to read their overflow flags back later)


class _Scene:
    """Everything the host remembers about ONE scene size (device, P, H, W) - the only two registries of this module are
    `_SCENES` (these) and `_CAMERAS` (per settings object, below):
      capacity     per-view pair-arena capacity learned so far (None: nothing learned yet; the first forward is checked)
      longest_bin  longest tile list a checked forward has reported (None: unknown) -> the sort hints of `_flags`
      tracks       camera set -> _AutoTrack of the "auto" sync mode (bounded: Topo4D builds new camera tuples every frame)
      plans        launch shape (V, M, degree, scale modifier) -> _Plan (ctypes structures, byte sizes per capacity)"""
    __slots__ = ("key", "capacity", "longest_bin", "tracks", "plans")
    MAX_TRACKS = 256

    def __init__(self, key):
        self.key = key
        self.capacity: Optional[int] = None
        self.longest_bin: Optional[int] = None
        self.tracks = {}
        self.plans = {}

    def grow(self, capacity: int) -> None:
        self.capacity = max(self.capacity or 0, int(capacity))

    def track_of(self, cam_key):
        """(track, is_new) of a camera set; the oldest half is forgotten when the table is full (dicts keep insertion order)."""
        t = self.tracks.get(cam_key)
        if t is not None:
            return t, False
        if len(self.tracks) >= self.MAX_TRACKS:
            for old in list(self.tracks)[: self.MAX_TRACKS // 2]:
                del self.tracks[old]
        t = self.tracks[cam_key] = _AutoTrack(self)
        return t, True


_SCENES = {}            # (device_index, P, H, W) -> _Scene


def _scene(device_index, P: int, H: int, W: int) -> _Scene:
    key = (device_index, int(P), int(H), int(W))
    sc = _SCENES.get(key)
    if sc is None:
        sc = _SCENES[key] = _Scene(key)
    return sc


def _forget_scenes() -> None:
    """Tests: drop everything learned about every scene size (capacities, sort hints, auto-mode tracks, pending statuses)."""
    _SCENES.clear()
    _PENDING.clear()


class _Pending:
    """One un-synchronised forward of the "auto" mode whose 16-byte binning status is still on its way to pinned host memory."""
    __slots__ = ("track", "slot", "cap", "done", "overflow", "need")

    def __init__(self, track, slot, cap):
        self.track, self.slot, self.cap = track, slot, cap
        self.done, self.overflow, self.need = False, False, 0

    def landed(self) -> bool:
        h = self.track.host
        return int(h[self.slot, 0]) != -1 and int(h[self.slot, 1]) != -1      # (both halves: a copy caught half-way reads as "not yet")


class _AutoTrack:
    """Bookkeeping of the "auto" sync mode for one (scene size, camera set): a ring of 16-byte status blocks in pinned host
    memory, one per un-synchronised forward still in flight, and the largest need seen.  A slot is filled with an impossible
    value before its forward is enqueued; the forward's asynchronous device-to-host copy overwrites it, so "has it landed?"
    is two host loads - no event, no stream object on the per-iteration path."""
    RING = 8
    __slots__ = ("pinned", "host", "args", "head", "count", "need", "scene", "live")

    def __init__(self, scene=None):
        self.scene = scene                              # the _Scene whose capacity this track feeds
        self.pinned = torch.zeros(self.RING, 2, dtype=torch.int64).pin_memory()
        self.host = self.pinned.numpy()                 # same memory, cheap scalar reads
        base = self.pinned.data_ptr()
        self.args = [C.cast(C.c_void_p(base + 16 * i), C.POINTER(T4DStatus)) for i in range(self.RING)]
        self.head = 0                                   # next slot to hand out
        self.count = 0                                  # slots in flight
        self.need = 0
        self.live = [None] * self.RING                  # per slot: the _Pending that owns it (statuses may be harvested out of order)

    def claim(self, cap: int) -> "_Pending":
        """Reserves the next slot for a forward that is about to be enqueued; `self.args[entry.slot]` is the status pointer to
        pass to it."""
        # the ring is full, or - after an out-of-order harvest (a backward waiting for ITS forward on another stream) - the slot
        # at the head still belongs to an older forward whose status the GPU has yet to write
        owner = self.live[self.head]
        if self.count == self.RING or (owner is not None and not owner.done):
            poll_truncation()
            owner = self.live[self.head]
            if self.count == self.RING or (owner is not None and not owner.done):
                dev_index = self.scene.key[0] if self.scene is not None else None
                torch.cuda.synchronize(dev_index)
                # everything that was enqueued on that device has landed by now: harvest it (a truncated pass among them raises
                # here); a slot that still reads "not yet" belongs to a forward that was never enqueued and can only be forgotten
                poll_truncation(_after_sync=True, _synced_device=dev_index)
        i = self.head
        self.host[i, 0] = -1
        self.host[i, 1] = -1
        self.head = (i + 1) % self.RING
        self.count += 1
        entry = _Pending(self, i, cap)
        self.live[i] = entry
        _PENDING.append(entry)
        return entry

    def unclaim(self, entry: "_Pending") -> None:
        """The forward this slot was claimed for failed before its status copy was enqueued: give the slot back."""
        if not entry.done:
            entry.done = True
            try:
                _PENDING.remove(entry)
            except ValueError:
                pass
            self.count -= 1
            if (entry.slot + 1) % self.RING == self.head:
                self.head = entry.slot


_PENDING = deque()      # _Pending entries of the un-synchronised forwards, in submission order


def _harvest(entry: "_Pending"):
    """Takes a landed status out of the books; returns (need, capacity used) if that forward was truncated."""
    track = entry.track
    raw0 = int(track.host[entry.slot, 0])
    entry.done = True
    track.count -= 1
    entry.overflow, entry.need = bool(raw0 & 0xffffffff), (raw0 >> 32) & 0xffffffff
    track.need = max(track.need, entry.need)
    if entry.overflow:
        if track.scene is not None:
            track.scene.grow(_round_capacity(track.need))
        return (entry.need, entry.cap)
    return None


def _truncation_error(truncated) -> RuntimeError:
    return RuntimeError(
        f"topo4d_amd (sync_mode='auto'): a render needed {truncated[0]} (Gaussian,tile) pairs per view but its arena held "
        f"{truncated[1]}; it was truncated (incomplete tile lists; its backward returns no gradients). The arena has been "
        "enlarged: re-run that iteration, or use set_sync_mode('checked') for scenes that change abruptly.")


def poll_truncation(wait_for: Optional["_Pending"] = None, _after_sync: bool = False, _synced_device=None) -> None:
    """"auto" sync mode: look at every binning status that has landed since the last look - of ALL camera sets, not only
    the one being rendered - grow the arenas they ask for, and raise RuntimeError if any of those forwards was truncated.
    Every auto-mode forward calls this first, every auto-mode BACKWARD calls it for its own forward once that status has landed
    (`wait_for`, below), and `FusedAdamPins.step()` calls it before it launches, so a truncated pass raises inside
    `loss.backward()` or, at the latest, before the fused optimiser step (a backward that ran ahead of its status carries zeros).
    Without `wait_for` it never synchronises: statuses are inspected in submission order, the first one still in flight ends
    the look.  With `wait_for` (the entry of one forward) it returns only once THAT status has landed: the copy was enqueued
    behind the forward's binning kernels, long before the loss and the backward were, so it has normally arrived already;
    if not, the host spins on the two pinned words (and falls back to a stream synchronisation after a second)."""
    truncated = None
    if wait_for is not None and not wait_for.done and not wait_for.landed():
        import time
        t0 = time.perf_counter()
        while not wait_for.landed():
            if time.perf_counter() - t0 > 1.0:
                # the device the forward ran on, not whichever is current
                sc = wait_for.track.scene
                torch.cuda.synchronize(sc.key[0] if sc is not None else None)
                if not wait_for.landed():
                    # (a forward on a capturing stream, or one that never ran): no gradient of a render whose status is unknown
                    raise RuntimeError("topo4d_amd (sync_mode='auto'): the binning status of this backward's forward has not arrived "
                                       "after a device synchronisation; refusing to return gradients of a render that may be truncated")
                break
    while _PENDING:
        entry = _PENDING[0]
        if not entry.landed():
            if _after_sync:
                sc = entry.track.scene
                if _synced_device is None or sc is None or sc.key[0] == _synced_device:
                    _PENDING.popleft()                  # can never land: its forward was not enqueued
                    entry.done = True
                    entry.track.count -= 1
                    continue
                # an entry of ANOTHER device: it may still be in flight - leave it, look past it
                later = [e for e in list(_PENDING)[1:] if e.landed()]
                for e in later:
                    _PENDING.remove(e)
                    truncated = _harvest(e) or truncated
            break
        _PENDING.popleft()
        truncated = _harvest(entry) or truncated
    if wait_for is not None and not wait_for.done and wait_for.landed():
        # an earlier status of ANOTHER stream is still in flight: take this one out of order
        try:
            _PENDING.remove(wait_for)
        except ValueError:
            pass
        truncated = _harvest(wait_for) or truncated
    if truncated is not None:
        raise _truncation_error(truncated)


def set_sync_mode(mode: str) -> None:
    """Without a call of this function `ViewBatch` / `rasterize_views` run "checked" and the one-view drop-in
    `GaussianRasterizer` - Topo4D's optimisation loop, one camera per call thousands of times per frame (train.py:661-673) -
    runs "auto" for calls that will be differentiated and "checked" otherwise (no_grad renders, `debug=True` cameras).
    After a call everything runs the mode it names.
    "checked": the forward synchronises once, after the binning sizes are known, and re-runs
    with a larger pair arena if it was too small — exactly where upstream reads `num_rendered` back.
    "lazy": never synchronises; tile lists are truncated (memory-safe) if the arena learned by earlier checked
    calls is too small, and `ViewBatch.fetch_status()` / `last_status()` reports it.  Use lazy only when the
    capacity was established by a checked call on (nearly) the same scene — bench.py does.
    "auto": for optimisation loops.  The first forward of every (scene size, camera set) is checked; afterwards the
    forward does not synchronise: the binning status is copied to pinned host memory asynchronously, right behind the binning
    kernels.  The arena is grown as soon as 75 % of it is in use (it is sized 1.5x the largest need seen), so consecutive
    iterations of an optimiser cannot overflow it.  Should a forward nevertheless be truncated (the scene jumped by more than
    a third between two calls), ITS OWN BACKWARD raises RuntimeError before it returns any gradient - `loss.backward()` fails,
    `optimizer.step()` is never reached - and the arena has been enlarged for the re-run.  The backward LOOKS at the pinned
    status words and never waits for them: they have normally landed long before (the loss sits between the forward and its
    backward); if they have not, the backward is launched, the library itself writes ZERO gradients for a truncated forward - no
    gradient of an incomplete render ever exists - and the RuntimeError comes from the next look: the next auto-mode forward,
    `FusedAdamPins.step()` (before it launches) or `poll_truncation()`."""
    global _SYNC_MODE, _SYNC_MODE_EXPLICIT
    if mode not in ("checked", "lazy", "auto"):
        raise ValueError("sync mode must be 'checked', 'lazy' or 'auto'")
    _SYNC_MODE = mode
    _SYNC_MODE_EXPLICIT = True


def get_sync_mode(drop_in: bool = False) -> str:
    """The mode `ViewBatch` / `rasterize_views` run (drop_in=False) or the one a differentiated call of the one-view
    `GaussianRasterizer` runs (drop_in=True: "auto" until set_sync_mode() names another)."""
    if drop_in and not _SYNC_MODE_EXPLICIT:
        return "auto"
    return _SYNC_MODE


def _save_sync_mode():
    """(mode, was it set explicitly) - for code that switches modes temporarily and must leave NO trace (loop.GraphedViews,
    bench.py): `set_sync_mode(get_sync_mode())` would turn the drop-in's default into an explicit choice for good."""
    return (_SYNC_MODE, _SYNC_MODE_EXPLICIT)


def _restore_sync_mode(saved) -> None:
    global _SYNC_MODE, _SYNC_MODE_EXPLICIT
    _SYNC_MODE, _SYNC_MODE_EXPLICIT = saved


def _initial_capacity(P: int) -> int:
    return max(16384, 8 * int(P))


def _round_capacity(n: int) -> int:
    n = int(n * 1.5) + 1024
    return (n + 1023) // 1024 * 1024


# ------------------------------------------------------------------------------------------------------------
# view records
# ------------------------------------------------------------------------------------------------------------
class _CameraRecord:
    """What the host keeps per `GaussianRasterizationSettings` object (train.py:98 builds each camera once per frame and reuses
    it for every iteration): the packed device record and - for the one-view drop-in - the call's constants."""
    __slots__ = ("settings", "version", "record", "spec")

    def __init__(self, settings, version, record):
        self.settings, self.version, self.record, self.spec = settings, version, record, None


_CAMERAS = OrderedDict()      # id(settings) -> _CameraRecord; least recently used first
_CAMERAS_MAX = 512


def _camera(s: GaussianRasterizationSettings, device, touch: bool = True) -> _CameraRecord:
    key = id(s)
    ver = (s.viewmatrix._version, s.projmatrix._version, s.campos._version, s.bg._version, device)
    hit = _CAMERAS.get(key)
    if hit is not None and hit.settings is s and hit.version == ver:
        if touch:
            _CAMERAS.move_to_end(key)                    # (one camera per call: the LRU order is refreshed on misses only)
        return hit
    rec = _CameraRecord(s, ver, _pack_one_view(s, device))
    _CAMERAS[key] = rec
    _CAMERAS.move_to_end(key)
    # evict the least recently used entries only: whoever still holds a record (a ViewBatch, a captured HIP graph via
    # loop.GraphedViews) keeps its own reference, so eviction can never free memory a pending launch reads
    while len(_CAMERAS) > _CAMERAS_MAX:
        _CAMERAS.popitem(last=False)
    return rec


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
    recs = [_camera(s, device, touch=len(settings) > 1).record for s in settings]
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


class _Plan:
    """Per launch shape of a scene size (`_Scene.plans`: V, M, degree, scale modifier) scratch of the host side: the ctypes structures of one call
    are built once and refilled (a call only uses them while the C function runs), and the byte sizes the library reports for
    a given pair capacity are remembered - Topo4D calls the rasterizer thousands of times per frame with the same shapes."""
    __slots__ = ("fio", "bio", "status", "state_bytes", "scratch_bytes", "fio_ref", "bio_ref", "status_ref")

    def __init__(self):
        self.fio, self.bio, self.status = T4DForwardIO(), T4DBackwardIO(), T4DStatus()
        self.fio_ref, self.bio_ref, self.status_ref = C.byref(self.fio), C.byref(self.bio), C.byref(self.status)
        self.state_bytes, self.scratch_bytes = {}, {}


_F32 = torch.float32


def _dp(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


class ViewBatch:
    """One forward (+ optional backward) of V views of the same Gaussians through the C ABI.

    Owns the opaque state buffer between forward and backward, like the (geomBuffer, binningBuffer, imgBuffer)
    byte tensors upstream keeps in its autograd ctx.

    `param_sets = S > 1`: the V views belong to S independent frames, V / S consecutive views each (T4DProblem.views_per_param_set);
    every per-Gaussian input of forward() then carries a leading set axis ([S,P,3] means3D, [S,P,1] opacities, ...), outputs and
    gradients stay per view.  A view-sharded rank renders its few cameras of several frames in one launch set this way.
    """

    def __init__(self, views: torch.Tensor, H: int, W: int, scale_modifier: float = 1.0, sh_degree: int = 0,
                 debug: bool = False, prefiltered: bool = False, cam_key=None, sync_mode: Optional[str] = None,
                 param_sets: int = 1):
        if not views.is_cuda:
            raise RuntimeError("topo4d_amd has no CPU path: tensors must live on a HIP device")
        self.lib = _lib.load()
        self.views = views if views.is_contiguous() else views.contiguous()
        self.V = int(views.shape[0])
        self.param_sets = int(param_sets)
        if self.param_sets < 1 or self.V % self.param_sets != 0:
            raise ValueError("the number of views must be a multiple of param_sets")
        self.H, self.W = int(H), int(W)
        self.scale_modifier = float(scale_modifier)
        self.sh_degree = int(sh_degree)
        self.debug = bool(debug)
        self.prefiltered = bool(prefiltered)
        self.device = views.device
        self.cam_key = cam_key if cam_key is not None else (views.data_ptr(), self.V)   # identity of the camera set ("auto" mode)
        self.sync_mode = sync_mode                      # None: the module-wide mode at the time of the forward
        self.state = None
        self.prob = None
        self.inputs = None
        self.radii = None
        self.plan = None
        self.pending = None
        self.flat_grads = False                         # True (autograd path, V = 1): gradients come back without the view axis
        # lazy mode only: 16 bytes of PINNED host memory (data pointer) that receive this forward's raw status block
        # { overflow | pairs needed << 32, total pairs } - a one-view launch's binning kernel writes them itself, so a captured
        # iteration (loop.GraphedViews) needs no copy node to keep the status of its forward
        self.status_sink = None
        # T4D_FLAG_RAW_PARAMS: `rotations`, `opacities`, `scales` of forward() are Topo4D's optimiser parameters (un-normalised
        # quaternions, logits, log scales - helpers.py:95-97); the library activates them itself and backward() returns the
        # gradients with respect to them
        self.raw_params = False

    # -- helpers ---------------------------------------------------------------------------------------------
    def _flags(self, P: int, checked: bool) -> int:
        flags = 0
        if checked:
            flags |= T4D_FLAG_CHECKED
        if self.debug:
            flags |= T4D_FLAG_DEBUG_SYNC
        if self.prefiltered:
            flags |= T4D_FLAG_PREFILTERED
        if self.raw_params:
            flags |= _lib.T4D_FLAG_RAW_PARAMS
        # tile lists of this scene size stayed well below the LDS sort buffer (2048) so far: skip the long-bin sort launch
        # (a speed hint only - see include/topo4d_raster.h)
        longest = _scene(self.device.index, P, self.H, self.W).longest_bin
        if longest is not None and longest > 1024:
            flags |= _lib.T4D_FLAG_LONG_LISTS     # some tile list bounds a small launch: the latency forward for up to 24 x CUs tiles
        if longest is not None and longest <= 1536:
            flags |= T4D_FLAG_NO_LONG_BINS
            if longest <= 500:            # ... and below the one-pass ranking sort (512; a bin that outgrows it is still sorted
                                          # correctly, by the slower LDS merge): small launches sort inside the render kernel
                flags |= _lib.T4D_FLAG_SHORT_BINS
        return flags

    def _stream(self):
        return _raw_stream(self.device)

    # -- forward ---------------------------------------------------------------------------------------------
    def forward(self, means3D, opacities, scales=None, rotations=None, colors_precomp=None, shs=None,
                cov3D_precomp=None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
        dev = self.device
        means3D = _f32c(means3D, "means3D", dev)
        if means3D is None:
            raise ValueError("means3D must not be empty")
        S = self.param_sets
        if S > 1 and (means3D.dim() != 3 or means3D.shape[0] != S):
            raise ValueError("param_sets > 1: means3D must be [S,P,3] (every per-Gaussian input carries the set axis)")
        P = int(means3D.shape[-2])
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
        if opacities is None or opacities.numel() != S * P:
            raise ValueError("opacities must hold one value per Gaussian")
        for name, t, width in (("scales", scales, 3), ("rotations", rotations, 4), ("colors_precomp", colors_precomp, 3),
                               ("cov3D_precomp", cov3D_precomp, 6)):
            if S > 1 and t is not None and t.numel() != S * P * width:
                raise ValueError(f"param_sets > 1: {name} must be [S,P,{width}]")
        M = 0
        if shs is not None:
            if shs.dim() != (3 if S == 1 else 4) or shs.shape[-3] != P or shs.shape[-1] != 3 or shs.numel() % (S * P * 3) != 0:
                raise ValueError("shs must be [P, M, 3]" if S == 1 else "param_sets > 1: shs must be [S, P, M, 3]")
            M = int(shs.shape[-2])
            if M < (self.sh_degree + 1) ** 2:
                raise ValueError("shs holds fewer coefficients than sh_degree needs")
        V, H, W = self.V, self.H, self.W
        if self.flat_grads:                       # the drop-in's own shapes (train.py:307): no view axis to strip afterwards
            color = torch.empty((3, H, W), dtype=_F32, device=dev)
            depth = torch.empty((1, H, W), dtype=_F32, device=dev)
            alpha = torch.empty((1, H, W), dtype=_F32, device=dev)
            radii = torch.empty((P,), dtype=torch.int32, device=dev)
        else:
            color = torch.empty((V, 3, H, W), dtype=_F32, device=dev)
            depth = torch.empty((V, 1, H, W), dtype=_F32, device=dev)
            alpha = torch.empty((V, 1, H, W), dtype=_F32, device=dev)
            radii = torch.empty((V, P), dtype=torch.int32, device=dev)

        mode = self.sync_mode or _SYNC_MODE
        scene = _scene(dev.index, P, H, W)
        known = scene.capacity is not None
        cap = scene.capacity if known else _initial_capacity(P)
        checked = (mode == "checked") or self.debug or not known
        track = None
        if mode == "auto" and not self.debug:
            if _PENDING:
                poll_truncation()                                 # statuses of every camera set that have landed by now
                cap = scene.capacity or cap
            track, new_track = scene.track_of(self.cam_key)
            if new_track:
                checked = True                                    # first time these cameras see this scene size
            else:
                if track.need > 0.75 * cap:                       # grow well before the arena can overflow
                    scene.grow(_round_capacity(track.need))
                cap = scene.capacity or cap
        pkey = (V, M, self.sh_degree, self.scale_modifier, S)
        plan = scene.plans.get(pkey)
        if plan is None:
            plan = scene.plans[pkey] = _Plan()
        self.plan = plan
        status = plan.status
        lib = self.lib
        stream = _raw_stream(dev)
        pending = None
        for _attempt in range(6):
            flags = self._flags(P, checked)
            status_arg = plan.status_ref
            if track is not None and not checked:
                flags |= _lib.T4D_FLAG_ASYNC_STATUS
                pending = track.claim(cap)
                status_arg = track.args[pending.slot]
            elif self.status_sink is not None and not checked:
                flags |= _lib.T4D_FLAG_ASYNC_STATUS
                status_arg = C.cast(C.c_void_p(int(self.status_sink)), C.POINTER(T4DStatus))
            prob = T4DProblem(T4D_ABI_VERSION, V, P, H, W, self.sh_degree, M, self.scale_modifier, cap, flags, 0 if S == 1 else V // S)
            nbytes = plan.state_bytes.get(cap)
            if nbytes is None:
                nbytes = plan.state_bytes[cap] = lib.t4d_state_bytes(C.byref(prob))
            state = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
            io = plan.fio
            io.views = self.views.data_ptr(); io.means3D = means3D.data_ptr(); io.opacities = opacities.data_ptr()
            io.scales = _dp(scales); io.rotations = _dp(rotations); io.cov3D_precomp = _dp(cov3D_precomp)
            io.colors_precomp = _dp(colors_precomp); io.shs = _dp(shs)
            io.out_color = color.data_ptr(); io.out_depth = depth.data_ptr(); io.out_alpha = alpha.data_ptr()
            io.out_radii = radii.data_ptr(); io.state = state.data_ptr(); io.state_bytes = nbytes
            rc = lib.t4d_rasterize_forward(C.byref(prob), plan.fio_ref, status_arg, stream)
            if rc == T4D_OK:
                break
            if pending is not None:                      # the status of a failed forward never lands: give its slot back
                track.unclaim(pending)
                pending = None
            if rc == T4D_ERR_PAIR_OVERFLOW:
                cap = _round_capacity(status.max_pairs_per_view)
                continue
            raise RuntimeError(f"t4d_rasterize_forward failed (code {rc}): {_lib.last_error()}")
        else:
            raise RuntimeError("pair arena kept overflowing; this should be impossible")
        if checked:
            # keep 1.5x head-room over what this scene needs so that lazy calls on nearby scenes fit
            want = _round_capacity(status.max_pairs_per_view)
            scene.capacity = max(want, cap if scene.capacity is not None else 0)
            scene.longest_bin = max(scene.longest_bin or 0, int(status.max_tile_pairs))
            if track is not None:
                track.need = max(track.need, int(status.max_pairs_per_view))
            st = T4DStatus(status.max_pairs_per_view, status.total_pairs, status.overflow, status.max_tile_pairs)
            self.last_status = st
        else:
            self.last_status = None
        self.prob, self.state, self.radii = prob, state, radii
        self.pending = pending                           # "auto" mode: this forward's status, looked at by its backward
        if _BATCH_LOG is not None:
            _BATCH_LOG.append(self)
        self.inputs = (means3D, opacities, scales, rotations, cov3D_precomp, colors_precomp, shs)
        return color, radii, depth, alpha

    # -- backward --------------------------------------------------------------------------------------------
    def backward(self, dL_dcolor, dL_ddepth=None, dL_dalpha=None, cotangent_dot=None):
        """Per-view gradients: dict of [V,P,...] tensors (means3D, means2D, colors_precomp|shs, opacities,
        scales+rotations|cov3D_precomp).  `cotangent_dot`: optional fp32 [V] tensor that receives
        <color, dL_dcolor> + <depth, dL_ddepth> + <alpha, dL_dalpha> per view (a by-product of the replay)."""
        if self.state is None:
            raise RuntimeError("backward() before forward()")
        if self.inputs is None:
            raise RuntimeError("this ViewBatch's inputs are owned by an autograd graph (GaussianRasterizer / rasterize_views): "
                               "call .backward() on the loss instead of ViewBatch.backward()")
        dev = self.device
        means3D, opacities, scales, rotations, cov3D_precomp, colors_precomp, shs = self.inputs
        V, P, H, W = self.V, int(means3D.shape[-2]), self.H, self.W
        M = 0 if shs is None else int(shs.shape[-2])
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
        lead = () if self.flat_grads else (V,)
        new = lambda *shape: torch.empty(lead + shape, dtype=_F32, device=dev)
        g = dict(means3D=new(P, 3), means2D=new(P, 3), opacities=new(P, 1))
        g["colors_precomp"] = new(P, 3) if colors_precomp is not None else None
        g["shs"] = new(P, M, 3) if shs is not None else None
        g["scales"] = new(P, 3) if cov3D_precomp is None else None
        g["rotations"] = new(P, 4) if cov3D_precomp is None else None
        g["cov3D_precomp"] = new(P, 6) if cov3D_precomp is not None else None
        prob, plan, lib = self.prob, self.plan, self.lib
        cap = prob.pair_capacity
        sbytes = plan.scratch_bytes.get(cap)
        if sbytes is None:
            sbytes = plan.scratch_bytes[cap] = lib.t4d_backward_scratch_bytes(C.byref(prob))
        scratch = torch.empty((sbytes,), dtype=torch.uint8, device=dev)
        io = plan.bio
        io.views = self.views.data_ptr(); io.means3D = means3D.data_ptr(); io.opacities = opacities.data_ptr()
        io.scales = _dp(scales); io.rotations = _dp(rotations); io.cov3D_precomp = _dp(cov3D_precomp)
        io.colors_precomp = _dp(colors_precomp); io.shs = _dp(shs)
        io.radii = self.radii.data_ptr(); io.state = self.state.data_ptr(); io.state_bytes = self.state.numel()
        io.dL_dcolor = dL_dcolor.data_ptr(); io.dL_ddepth = _dp(dL_ddepth); io.dL_dalpha = _dp(dL_dalpha)
        io.dL_dmeans3D = g["means3D"].data_ptr(); io.dL_dmeans2D = g["means2D"].data_ptr()
        io.dL_dcolors = _dp(g["colors_precomp"]); io.dL_dshs = _dp(g["shs"]); io.dL_dopacities = g["opacities"].data_ptr()
        io.dL_dscales = _dp(g["scales"]); io.dL_drotations = _dp(g["rotations"]); io.dL_dcov3D = _dp(g["cov3D_precomp"])
        io.scratch = scratch.data_ptr(); io.scratch_bytes = sbytes; io.cotangent_dot = _dp(cotangent_dot)
        pending = self.pending
        if pending is not None:
            # "auto" mode: no gradient of a truncated render ever leaves this function.  The status was copied out right behind
            # the forward's binning kernels and is LOOKED AT here, after the host-side preparation above - two host loads, never a
            # wait.  Landed (the normal case: the loss was enqueued in between): a truncated forward raises now, before any gradient
            # exists.  Not landed yet (a host running ahead of the device): the backward is launched all the same - the library
            # writes ZERO gradients for a truncated forward (include/topo4d_raster.h) - and the truncation raises at the next look
            # (every auto-mode forward, FusedAdamPins.step(), poll_truncation()).  Waiting here instead made host and device hand
            # over to each other once per iteration: two throughput regimes of one build, 4-5 k and 8-9 k it/s (round 5).
            if not pending.done and pending.landed():
                poll_truncation(wait_for=pending)
            if pending.done and pending.overflow:
                raise _truncation_error((pending.need, pending.cap))
        rc = lib.t4d_rasterize_backward(C.byref(prob), plan.bio_ref, _raw_stream(dev))
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
class _CallSpec:
    """Everything of a render call that is not a differentiable tensor, as ONE autograd-invisible argument."""
    __slots__ = ("views", "H", "W", "scale_modifier", "sh_degree", "debug", "prefiltered", "cam_key", "sync_mode", "one_view")

    def __init__(self, views, H, W, scale_modifier, sh_degree, debug, prefiltered, cam_key, sync_mode=None, one_view=False):
        self.views, self.H, self.W, self.scale_modifier, self.sh_degree = views, H, W, scale_modifier, sh_degree
        self.debug, self.prefiltered, self.cam_key, self.sync_mode, self.one_view = debug, prefiltered, cam_key, sync_mode, one_view


class _RasterizeViews(torch.autograd.Function):
    """Inputs in upstream's order (None where upstream passes an empty tensor) + the call's _CallSpec; outputs
    (color[V,3,H,W], radii[V,P], depth[V,1,H,W], alpha[V,1,H,W]), or without the view axis for the one-view drop-in."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, spec):
        batch = ViewBatch(spec.views, spec.H, spec.W, spec.scale_modifier, spec.sh_degree, spec.debug, spec.prefiltered,
                          spec.cam_key, spec.sync_mode)
        batch.flat_grads = spec.one_view
        color, radii, depth, alpha = batch.forward(means3D, opacities, scales, rotations, colors_precomp, sh, cov3Ds_precomp)
        ctx.batch = batch
        # the backward kernels re-read the forward inputs: saving them through autograd (as upstream does) makes an in-place
        # update between forward and backward raise instead of silently differentiating at the wrong point
        inputs = batch.inputs
        ctx.input_slots = slots = [i for i in range(7) if inputs[i] is not None]
        ctx.save_for_backward(*[inputs[i] for i in slots])
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
        batch.inputs = inputs
        if grad_color is None:
            grad_color = torch.zeros(batch.V, 3, batch.H, batch.W, dtype=torch.float32, device=batch.device)
        g = batch.backward(grad_color, grad_depth, grad_alpha)
        batch.inputs = None                  # the batch (kept by GraphedViews / _BATCH_LOG) must not keep the inputs alive
        ctx.batch = None
        shp3, shp2, shpo = ctx.shapes
        if batch.flat_grads:                 # one view: the kernel's outputs already have the inputs' shapes
            d2 = g["means2D"] if (shp2 is not None and ctx.needs_input_grad[1]) else None
            go = g["opacities"]
            return (g["means3D"], d2, g["shs"], g["colors_precomp"], go if go.shape == shpo else go.reshape(shpo),
                    g["scales"], g["rotations"], g["cov3D_precomp"], None)
        if batch.V > 1:
            g = _sum_views(g, batch.V, need_means2D=shp2 is not None and ctx.needs_input_grad[1])
            red = lambda t: t                # already summed over the views, in ONE launch (t4d_sum_views)
        else:
            red = lambda t: None if t is None else t[0]
        d_means2D = None
        if shp2 is not None and ctx.needs_input_grad[1]:
            d_means2D = red(g["means2D"]).reshape(shp2)
        return (red(g["means3D"]).reshape(shp3), d_means2D, red(g["shs"]), red(g["colors_precomp"]),
                red(g["opacities"]).reshape(shpo), red(g["scales"]), red(g["rotations"]), red(g["cov3D_precomp"]), None)


def _sum_views(g, V: int, need_means2D: bool):
    """dict of per-view gradients [V,P,...] -> dict of view-summed gradients [P,...] (None stays None), one launch."""
    lib = _lib.load()
    names = [k for k, t in g.items() if t is not None and (k != "means2D" or need_means2D)]
    out = {k: None for k in g}
    if not names:
        return out
    for k in names:
        out[k] = torch.empty(g[k].shape[1:], dtype=_F32, device=g[k].device)
    n = len(names)
    src = (C.c_void_p * n)(*[g[k].data_ptr() for k in names])
    dst = (C.c_void_p * n)(*[out[k].data_ptr() for k in names])
    cnt = (C.c_int64 * n)(*[out[k].numel() for k in names])
    rc = lib.t4d_sum_views(V, n, src, dst, cnt, _raw_stream(g[names[0]].device))
    if rc != T4D_OK:
        raise RuntimeError(f"t4d_sum_views failed (code {rc}): {_lib.last_error()}")
    return out


def _none_if_empty(t):
    return None if (t is None or t.numel() == 0) else t


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
    debug = any(bool(s.debug) for s in settings)
    pref = any(bool(s.prefiltered) for s in settings)
    cam_key = tuple(id(s_) for s_ in settings)           # Settings tuples are built once per camera per frame (train.py:98)
    spec = _CallSpec(views, H, W, smod, deg, debug, pref, cam_key)
    return _RasterizeViews.apply(means3D, means2D, _none_if_empty(shs), _none_if_empty(colors_precomp), opacities,
                                 _none_if_empty(scales), _none_if_empty(rotations), _none_if_empty(cov3D_precomp), spec)


def _spec_of(s: GaussianRasterizationSettings, device) -> _CallSpec:
    """The one-view drop-in's per-camera constants, kept in the camera's record."""
    rec = _camera(s, device, touch=False)
    if rec.spec is None:
        rec.spec = _CallSpec(rec.record.unsqueeze(0), int(s.image_height), int(s.image_width), float(s.scale_modifier), int(s.sh_degree),
                             bool(s.debug), bool(s.prefiltered), (id(s),), None, True)
    return rec.spec


try:
    from torch.nn.modules.module import _has_any_global_hook as _has_global_hooks
except ImportError:                                          # pragma: no cover - torch without the helper: take the slow path
    def _has_global_hooks() -> bool:
        return True


class GaussianRasterizer(nn.Module):
    """Drop-in for `diff_gaussian_rasterization.GaussianRasterizer` (constructed per call at train.py:307).

    Topo4D builds one of these per ITERATION, so construction is on the hot path: the nn.Module machinery (a dozen dicts) is
    set up lazily, the first time anything asks for it (`.to()`, `.parameters()`, hooks ...), and calling the object goes
    straight to `forward` (nn.Module.__call__ only adds the hook dispatch, and a freshly built module has no hooks)."""

    def __init__(self, raster_settings: GaussianRasterizationSettings):
        object.__setattr__(self, "raster_settings", raster_settings)

    def _module_state(self) -> None:
        if "_modules" not in self.__dict__:                 # first use of the nn.Module state
            nn.Module.__init__(self)

    def __getattr__(self, name):
        if "_modules" not in self.__dict__:
            nn.Module.__init__(self)
            return getattr(self, name)
        return nn.Module.__getattr__(self, name)

    def __setattr__(self, name, value):                     # (assigning a submodule / parameter / flag to a fresh instance)
        self._module_state()
        nn.Module.__setattr__(self, name, value)

    def train(self, mode: bool = True):                     # .eval() on a fresh instance must not be undone by the lazy set-up
        self._module_state()
        return nn.Module.train(self, mode)

    def __call__(self, *args, **kwargs):
        if ("_forward_hooks" in self.__dict__ and (self._forward_hooks or self._forward_pre_hooks)) or _has_global_hooks():
            self._module_state()
            return nn.Module.__call__(self, *args, **kwargs)
        return self.forward(*args, **kwargs)

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
        if not means3D.is_cuda:
            raise RuntimeError("topo4d_amd has no CPU path: tensors must live on a HIP device")
        spec = _spec_of(self.raster_settings, means3D.device)
        if _SYNC_MODE_EXPLICIT:
            spec.sync_mode = None            # whatever set_sync_mode() named
        elif torch.is_grad_enabled() and (means3D.requires_grad or (colors_precomp is not None and colors_precomp.requires_grad)
                                          or opacities.requires_grad or (scales is not None and scales.requires_grad)
                                          or (rotations is not None and rotations.requires_grad)
                                          or (shs is not None and shs.requires_grad)
                                          or (cov3D_precomp is not None and cov3D_precomp.requires_grad)):
            # the optimisation-loop default of the one-view drop-in: un-synchronised, and the call's own backward refuses to
            # return gradients of a truncated render (see set_sync_mode)
            spec.sync_mode = "auto"
        else:
            spec.sync_mode = "checked"       # nothing will be differentiated (evaluation renders): upstream's own behaviour
        # (color[3,H,W], radii[P], depth[1,H,W], alpha[1,H,W]) - the four tensors train.py:307 unpacks
        return _RasterizeViews.apply(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, spec)
