"""Device-side handle: a thin wrapper over the C ABI of include/ptcore.h.

All compute happens in libptcore.so (hand-written HIP for gfx950). There is no CPU
fallback: constructing a Renderer without the library or without a GPU raises.
"""
import ctypes as C

import numpy as np

from . import _ffi


class CoreError(RuntimeError):
    pass


def _check(rc):
    if rc:
        raise CoreError(_ffi.core_lib().pt_last_error().decode("utf-8", "replace"))


CONTEXTS_CREATED = 0  # > 0: this process has initialised the GPU (tests that spawn the CLI look at it)


class Renderer:
    """One context = one GPU. Mirrors the device half of src/main.cu:74-111."""

    def __init__(self, device=0, _borrowed=None):
        self._lib = _ffi.core_lib()
        self._ctx = C.c_void_p()
        global CONTEXTS_CREATED
        CONTEXTS_CREATED += 1
        self._owned = _borrowed is None
        if _borrowed is None:
            _check(self._lib.pt_create(int(device), C.byref(self._ctx)))
        else:
            self._ctx = C.c_void_p(_borrowed)  # a context that belongs to a Group
        self.device = int(device)
        self.width = self.height = 0
        self._keep = []

    # -- uploads -----------------------------------------------------------------------
    def upload(self, scene, bvh, lst):
        """d_scene / d_bvh / d_lst .device_from_host (src/main.cu:74-76)."""
        _check(self._lib.pt_scene_upload(self._ctx, C.byref(scene.desc)))
        nodes = np.ascontiguousarray(bvh.nodes)
        indices = np.ascontiguousarray(bvh.indices)
        _check(self._lib.pt_bvh_upload(self._ctx, nodes.ctypes.data, len(nodes), indices.ctypes.data, len(indices)))
        ln = np.ascontiguousarray(lst.nodes)
        _check(self._lib.pt_lst_upload(self._ctx, ln.ctypes.data if len(ln) else None, len(ln)))

    def configure(self, width, height, exposure=0.0, clear_color=(0.0, 0.0, 0.0)):
        cfg = _ffi.RenderConfig(int(width), int(height), float(exposure), _ffi.Vec3(*[float(c) for c in clear_color]))
        _check(self._lib.pt_configure(self._ctx, C.byref(cfg)))
        self.width, self.height = int(width), int(height)

    # -- rendering ---------------------------------------------------------------------
    def render_batch(self, seed, previous_samples, current_samples):
        """One blocking launch into the library-owned accumulator (src/main.cu:104-106)."""
        _check(self._lib.pt_render_batch(self._ctx, int(seed), int(previous_samples), int(current_samples)))

    @staticmethod
    def _stream_arg(stream):
        """None = the context's own stream. A handle of 0 is the legacy default stream (torch's default
        stream reports 0): in the C ABI a null stream means "the context's own", which is NOT ordered against
        the default stream, so passing 0 here would silently launch unordered. Refused: use a
        torch.cuda.Stream (its .cuda_stream is non-zero), or stream=None plus synchronize()."""
        if stream is None:
            return None
        if int(stream) == 0:
            raise ValueError("stream handle 0 (the default stream) cannot be passed through the C ABI: "
                             "use a torch.cuda.Stream, or stream=None and Renderer.synchronize()")
        return C.c_void_p(int(stream))

    def render_batch_into(self, device_ptr, seed, previous_samples, current_samples, stream=None):
        """Asynchronous launch into a caller-owned device buffer (data_ptr of a torch tensor)."""
        _check(self._lib.pt_render_batch_into(self._ctx, C.c_void_p(int(device_ptr)), self._stream_arg(stream),
                                              int(seed), int(previous_samples), int(current_samples)))

    def render_batch_static(self, device_ptr, width, height, seed, previous_samples, current_samples, stream=None,
                            exposure=0.0, clear_color=(0.0, 0.0, 0.0)):
        """pt_render_batch_static: the same launch with its tile scratch in the library's static device
        array (no allocation, no host block); image size, exposure and clear colour come with the call."""
        cfg = _ffi.RenderConfig(int(width), int(height), float(exposure), _ffi.Vec3(*[float(c) for c in clear_color]))
        _check(self._lib.pt_render_batch_static(self._ctx, C.c_void_p(int(device_ptr)), self._stream_arg(stream), C.byref(cfg),
                                                int(seed), int(previous_samples), int(current_samples)))

    def render(self, seed, samples, samples_every_update):
        """The batch loop of src/main.cu:101-133 without the per-batch PNG: seed+1 per batch."""
        self.clear()
        done = 0
        while done < samples:
            cur = min(samples_every_update, samples - done)
            self.render_batch(seed, done, cur)
            done += cur
            seed += 1
        return self.read_image()

    def read_image(self):
        img = np.empty((self.height, self.width, 3), np.float32)
        _check(self._lib.pt_read_image(self._ctx, img.ctypes.data))
        return img

    def clear(self):
        _check(self._lib.pt_clear_image(self._ctx))

    def synchronize(self):
        _check(self._lib.pt_synchronize(self._ctx))

    # -- verification / measurement ------------------------------------------------------
    def rng_stream(self, seed, subsequence, n):
        out = np.zeros(n, np.uint32)
        _check(self._lib.pt_rng_stream(self._ctx, seed & 0xFFFFFFFFFFFFFFFF, subsequence, out.ctypes.data, n))
        return out

    def primary_hits(self, seed, jitter=True, tree=0, with_ties=False):
        """Face id and distance of every pixel's first camera ray, by the product walker on `tree`
        (2 binary, 4 four-wide, 0 the render launches' choice). with_ties: also the number of rays that were
        walked again sequentially because two candidates lay within rounding of each other."""
        ids = np.zeros((self.height, self.width), np.int32)
        dist = np.zeros((self.height, self.width), np.float32)
        redone = C.c_uint64(0)
        _check(self._lib.pt_primary_hits(self._ctx, int(seed), 1 if jitter else 0, int(tree), ids.ctypes.data, dist.ctypes.data,
                                         C.byref(redone)))
        return (ids, dist, int(redone.value)) if with_ties else (ids, dist)

    def count_batch(self, seed, current_samples):
        c = _ffi.CoreCounters()
        _check(self._lib.pt_count_batch(self._ctx, int(seed), int(current_samples), C.byref(c)))
        return c.as_dict()

    def last_kernel_ms(self):
        return float(self._lib.pt_last_kernel_ms(self._ctx))

    def kernel_ms(self, launches_back=0):
        """HIP-event duration of an earlier render launch (0 = the most recent one)."""
        return float(self._lib.pt_kernel_ms(self._ctx, int(launches_back)))

    ARITHMETIC = {"exact": 0, "fast": 1}

    def set_arithmetic(self, name):
        """"exact" (default: bit-identical to the CPU oracle) or "fast" (tolerance arithmetic), include/ptcore.h."""
        _check(self._lib.pt_set_arithmetic(self._ctx, self.ARITHMETIC[name]))

    def arithmetic(self):
        return "fast" if int(self._lib.pt_arithmetic(self._ctx)) == 1 else "exact"

    def walk_tree(self):
        """2 = binary tree, 4 = four-wide, 0 = not chosen yet (see include/ptcore.h)."""
        return int(self._lib.pt_walk_tree(self._ctx))

    def device_footprint(self):
        return int(self._lib.pt_device_footprint(self._ctx))

    def close(self):
        if self._ctx and self._owned:
            self._lib.pt_destroy(self._ctx)
        self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def group_plan(seed, samples, samples_every_update, ranks):
    """pt_group_plan: ([(index, rank, seed, previous_samples, current_samples)], [weight per rank]). Pure host
    arithmetic: needs the library, not a GPU."""
    lib = _ffi.core_lib()
    n = lib.pt_group_plan(int(seed), int(samples), int(samples_every_update), int(ranks), None, 0, None)
    if n < 0:
        raise CoreError(lib.pt_last_error().decode("utf-8", "replace"))
    batches = (_ffi.GroupBatch * n)()
    weights = (C.c_float * int(ranks))()
    lib.pt_group_plan(int(seed), int(samples), int(samples_every_update), int(ranks), batches, n, weights)
    return ([(b.index, b.rank, b.seed, b.previous_samples, b.current_samples) for b in batches], [float(w) for w in weights])


class Group:
    """pt_group: N GPUs in one process, batches sharded by index, one RCCL sum-reduce to rank 0
    (include/ptcore.h, "multi-GPU in one process")."""

    def __init__(self, devices):
        self._lib = _ffi.core_lib()
        self._g = C.c_void_p()
        global CONTEXTS_CREATED
        CONTEXTS_CREATED += 1
        arr = (C.c_int * len(devices))(*[int(d) for d in devices])
        _check(self._lib.pt_group_create(arr, len(devices), C.byref(self._g)))
        self.ranks = [Renderer(d, _borrowed=self._lib.pt_group_context(self._g, r)) for r, d in enumerate(devices)]

    def upload(self, scene, bvh, lst):
        for r in self.ranks:
            r.upload(scene, bvh, lst)

    def configure(self, width, height, exposure=0.0, clear_color=(0.0, 0.0, 0.0)):
        for r in self.ranks:
            r.configure(width, height, exposure, clear_color)

    def render(self, seed, samples, samples_every_update):
        _check(self._lib.pt_group_render(self._g, int(seed), int(samples), int(samples_every_update)))
        return self.ranks[0].read_image()

    def close(self):
        if self._g:
            for r in self.ranks:
                r.close()
            self._lib.pt_group_destroy(self._g)
            self._g = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
