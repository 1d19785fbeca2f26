"""ntt_amd -- MI355X-native NTT polynomial multiplication.

Thin ctypes binding over the C ABI of ``libntt_amd.so`` (``include/ntt_amd.h``); all compute is
hand-written HIP for gfx950 inside that library.  There is no CPU fallback: if the shared library
is missing this module raises at import, and every compute call fails without a gfx950 device.

The Python surface mirrors the reference harness (``main-montgomery.cc``):

* ``poly_multiply(a, b, p)``                 <- ``poly_multiply(int*a,int*b,int*ab,int n,int p)``
* ``read_case / check_case / write_case``    <- ``fRead / fCheck / fWrite``
* ``Context`` / ``Plan``                     <- the handle API for batched, device-resident work

Arrays may be numpy arrays (host memory; the library stages them) or objects exposing
``data_ptr()`` (torch CUDA tensors: used in place on the context's stream).
"""
import ctypes as C
import importlib.util
import os
import weakref

import numpy as np

# PyTorch-ROCm bundles its own HIP + HSA runtimes.  Two HSA runtimes cannot drive the GPU from one
# process; if torch is loaded AFTER this library it brings a second one and torch.cuda then reports
# "No HIP GPUs are available".  Loaded first, torch's HSA runtime (same soname) is the one the
# system HIP runtime under libntt_amd.so binds to, and both work side by side.  So: torch first.
if importlib.util.find_spec("torch") is not None:
    import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# NTT_AMD_LIB selects another build of the same library (A/B experiments under ntt_amd/variants/)
_LIB_PATH = os.environ.get("NTT_AMD_LIB") or os.path.join(_HERE, "libntt_amd.so")

OK, ERR_INVALID, ERR_UNSUPPORTED, ERR_HIP, ERR_NOMEM, ERR_IO, ERR_MISMATCH, ERR_INTERNAL = range(8)
MONTGOMERY, BARRETT = 0, 1

EXPORTED_SYMBOLS = (
    "poly_multiply", "ntt_ctx_create", "ntt_ctx_destroy", "ntt_ctx_set_stream", "ntt_ctx_stream",
    "ntt_ctx_synchronize", "ntt_plan_create", "ntt_plan_destroy", "ntt_plan_length",
    "ntt_poly_multiply_batched", "ntt_poly_multiply_batched_noalloc", "ntt_forward", "ntt_inverse", "ntt_crt_multiply", "ntt_crt_multiply_batched", "ntt_crt_prime",
    "ntt_negacyclic_plan_create", "ntt_negacyclic_multiply_batched",
    "ntt_negacyclic_plan64_create", "ntt_negacyclic_multiply64_batched",
    "ntt_plan64_create", "ntt_plan64_destroy", "ntt_plan64_length", "ntt_poly_multiply64_batched",
    "ntt_crt_residue_multiply", "ntt_crt_merge", "ntt_dist_plan_create", "ntt_dist_plan_destroy",
    "ntt_dist_colpass", "ntt_dist_local", "ntt_dist_local_length",
    "ntt_ctx_device_count", "ntt_ctx_device", "ntt_partition", "ntt_poly_multiply_batched_slots", "ntt_poly_multiply_limbs", "ntt_dist_forward", "ntt_dist_inverse", "ntt_dist_plan_transport", "ntt_dist_multiply",
    "ntt_comm_rccl_id", "ntt_comm_create_rccl", "ntt_comm_create_host", "ntt_comm_destroy", "ntt_comm_size", "ntt_comm_dist_forward", "ntt_comm_dist_inverse", "ntt_comm_dist_multiply", "ntt_comm_crt_multiply",
    "ntt_ctx_enable_kernel_timing", "ntt_ctx_kernel_timing_select", "ntt_ctx_kernel_timing_stride", "ntt_ctx_kernel_time", "ntt_last_error", "ntt_read_case",
    "ntt_check_case", "ntt_write_case", "ntt_poly_multiply_stream", "ntt_host_alloc", "ntt_host_free",
    "ntt_pack_case", "ntt_read_case_bin", "ntt_check_case_bin", "ntt_write_case_bin",
)


class NttError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"ntt_amd error {code}: {msg}")
        self.code = code


# ntt_alltoall_fn (include/ntt_amd.h): int (*)(void *user, const void *send, void *recv, size_t bytes_per_rank)
ALLTOALL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)
COMM_ID_BYTES = 128


def _load():
    if not os.path.exists(_LIB_PATH):
        raise ImportError(
            f"{_LIB_PATH} is missing: build it with `make -C ntt_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
    lib = C.CDLL(_LIB_PATH)
    vp, u32, u64, i32, sz = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int, C.c_size_t
    lib.poly_multiply.argtypes = [vp, vp, vp, i32, i32]
    lib.poly_multiply.restype = None
    lib.ntt_ctx_create.argtypes = [C.POINTER(i32), i32, C.POINTER(vp)]
    lib.ntt_ctx_destroy.argtypes = [vp]
    lib.ntt_ctx_set_stream.argtypes = [vp, vp]
    lib.ntt_ctx_stream.argtypes = [vp]
    lib.ntt_ctx_stream.restype = vp
    lib.ntt_ctx_synchronize.argtypes = [vp]
    lib.ntt_plan_create.argtypes = [vp, u32, i32, i32, i32, C.POINTER(vp)]
    lib.ntt_plan_destroy.argtypes = [vp]
    lib.ntt_plan_length.argtypes = [vp]
    lib.ntt_poly_multiply_batched.argtypes = [vp, vp, vp, vp, sz, i32]
    lib.ntt_poly_multiply_batched_noalloc.argtypes = [vp, vp, vp, vp, sz, i32]
    lib.ntt_forward.argtypes = [vp, vp, i32]
    lib.ntt_inverse.argtypes = [vp, vp, i32]
    lib.ntt_crt_multiply.argtypes = [vp, vp, vp, vp, i32, u64]
    lib.ntt_crt_multiply_batched.argtypes = [vp, vp, vp, vp, i32, u64, i32]
    lib.ntt_crt_prime.argtypes = [i32, C.POINTER(u32)]
    lib.ntt_negacyclic_plan_create.argtypes = [vp, u32, i32, i32, i32, C.POINTER(vp)]
    lib.ntt_negacyclic_multiply_batched.argtypes = [vp, vp, vp, vp, sz, i32]
    lib.ntt_plan64_create.argtypes = [vp, u64, i32, i32, C.POINTER(vp)]
    lib.ntt_negacyclic_plan64_create.argtypes = [vp, u64, i32, i32, C.POINTER(vp)]
    lib.ntt_negacyclic_multiply64_batched.argtypes = [vp, vp, vp, vp, sz, i32]
    lib.ntt_plan64_destroy.argtypes = [vp]
    lib.ntt_plan64_length.argtypes = [vp, C.POINTER(u32)]
    lib.ntt_poly_multiply64_batched.argtypes = [vp, vp, vp, vp, sz, i32]
    lib.ntt_crt_residue_multiply.argtypes = [vp, vp, vp, vp, i32, i32]
    lib.ntt_crt_merge.argtypes = [vp, vp, vp, vp, vp, vp, sz, u64]
    lib.ntt_dist_plan_create.argtypes = [vp, u32, i32, i32, i32, i32, C.POINTER(vp)]
    lib.ntt_dist_plan_destroy.argtypes = [vp]
    lib.ntt_dist_colpass.argtypes = [vp, vp, i32]
    lib.ntt_dist_local.argtypes = [vp, vp, i32]
    lib.ntt_dist_local_length.argtypes = [vp]
    lib.ntt_ctx_device_count.argtypes = [vp]
    lib.ntt_partition.argtypes = [C.c_long, i32, i32, C.POINTER(C.c_long), C.POINTER(C.c_long)]
    lib.ntt_poly_multiply_limbs.argtypes = [C.POINTER(vp), i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), sz, i32]
    lib.ntt_poly_multiply_batched_slots.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), sz, C.POINTER(i32)]
    lib.ntt_ctx_device.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(i32)]
    lib.ntt_dist_forward.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
    lib.ntt_dist_inverse.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
    lib.ntt_dist_plan_transport.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    lib.ntt_dist_multiply.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    lib.ntt_comm_rccl_id.argtypes = [vp]
    lib.ntt_comm_create_rccl.argtypes = [vp, vp, i32, i32, C.POINTER(vp)]
    lib.ntt_comm_create_host.argtypes = [vp, ALLTOALL_FN, vp, i32, i32, C.POINTER(vp)]
    lib.ntt_comm_destroy.argtypes = [vp]
    lib.ntt_comm_size.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    lib.ntt_comm_dist_forward.argtypes = [vp, vp, vp, vp]
    lib.ntt_comm_dist_inverse.argtypes = [vp, vp, vp, vp]
    lib.ntt_comm_crt_multiply.argtypes = [vp, vp, vp, vp, i32, u64]
    lib.ntt_comm_dist_multiply.argtypes = [vp, vp, vp, vp, vp]
    lib.ntt_ctx_enable_kernel_timing.argtypes = [vp, i32]
    lib.ntt_ctx_kernel_timing_select.argtypes = [vp, i32]
    lib.ntt_ctx_kernel_timing_stride.argtypes = [vp, i32]
    lib.ntt_ctx_kernel_time.argtypes = [vp, i32, C.POINTER(C.c_char_p), C.POINTER(C.c_double),
                                        C.POINTER(C.c_long)]
    lib.ntt_last_error.restype = C.c_char_p
    lib.ntt_read_case.argtypes = [C.c_char_p, i32, vp, vp, sz, C.POINTER(i32), C.POINTER(u64)]
    lib.ntt_check_case.argtypes = [C.c_char_p, i32, vp, i32]
    lib.ntt_write_case.argtypes = [C.c_char_p, i32, vp, i32]
    lib.ntt_poly_multiply_stream.argtypes = [vp, vp, vp, vp, sz, sz]
    lib.ntt_host_alloc.argtypes = [sz, C.POINTER(vp)]
    lib.ntt_host_free.argtypes = [vp]
    lib.ntt_pack_case.argtypes = [C.c_char_p, i32]
    lib.ntt_read_case_bin.argtypes = [C.c_char_p, i32, vp, vp, sz, C.POINTER(i32), C.POINTER(u64), C.POINTER(i32)]
    lib.ntt_check_case_bin.argtypes = [C.c_char_p, i32, vp, i32, i32]
    lib.ntt_write_case_bin.argtypes = [C.c_char_p, i32, vp, i32, u64, i32]
    return lib


_lib = _load()


def _check(rc):
    if rc != OK:
        raise NttError(rc, _lib.ntt_last_error().decode("utf-8", "replace"))


def _ptr(x, dtype, need=0):
    """Raw address of a numpy array (host) or of an object with data_ptr() (device tensor), after checking
    on the host what the kernels assume: element width, contiguity and at least `need` elements."""
    width = np.dtype(dtype).itemsize
    if hasattr(x, "data_ptr"):
        if x.element_size() != width or not x.is_contiguous():
            raise TypeError(f"expected a contiguous tensor with {width}-byte elements")
        have = x.numel()
        addr = x.data_ptr()
    else:
        if not isinstance(x, np.ndarray) or x.dtype != dtype or not x.flags["C_CONTIGUOUS"]:
            raise TypeError(f"expected a C-contiguous numpy array of {np.dtype(dtype)} or a device tensor")
        have = x.size
        addr = x.ctypes.data
    if have < need:
        raise ValueError(f"array of {have} elements where the call reads or writes {need}")
    return C.c_void_p(addr)


class Context:
    """One GPU + one stream (``ntt_ctx``), or -- ``Context(devices=[0, 1, 2, 3])`` -- a multi-device context with one
    slot per listed device (ids may repeat: a one-GPU box rehearses ``[0, 0, 0, 0]``), whose plans, products and
    transforms are sharded over the slots inside the library (``include/ntt_amd.h``, ``ntt_ctx_create``)."""

    def __init__(self, device=0, devices=None):
        h = C.c_void_p()
        if devices is None:
            dev = C.c_int(device)
            _check(_lib.ntt_ctx_create(C.byref(dev), 1, C.byref(h)))
            self.devices = [device]
        else:
            self.devices = [int(d) for d in devices]
            arr = (C.c_int * len(self.devices))(*self.devices)
            _check(_lib.ntt_ctx_create(arr, len(self.devices), C.byref(h)))
            device = self.devices[0]
        self._h = h
        self.device = device
        self._plans = weakref.WeakSet()      # live plans: a plan's destroy call dereferences its context

    @property
    def device_count(self):
        return _lib.ntt_ctx_device_count(self._h)

    def slot(self, index):
        """The single-device context of slot `index` (owned by this context; do not close it)."""
        h, dev = C.c_void_p(), C.c_int()
        _check(_lib.ntt_ctx_device(self._h, int(index), C.byref(h), C.byref(dev)))
        view = Context.__new__(Context)
        view._h, view.device, view.devices, view._plans, view._borrowed = h, dev.value, [dev.value], weakref.WeakSet(), self
        self._views = getattr(self, "_views", [])
        self._views.append(weakref.ref(view))           # its plans are closed with this context (they point into it)
        return view

    def close(self):
        """Destroys the context after closing every plan made from it that is still open (a plan handle
        holds a pointer to its context: destroying it later would be a use-after-free in native code)."""
        if self._h:
            for ref in getattr(self, "_views", []):           # slot views handed out by slot(): their plans first
                view = ref()
                if view is not None:
                    view.close()
            for pl in list(self._plans):
                pl.close()
            if getattr(self, "_borrowed", None) is None:      # a slot view does not own its handle
                _lib.ntt_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_ptr):
        _check(_lib.ntt_ctx_set_stream(self._h, C.c_void_p(stream_ptr)))

    def use_torch_stream(self):
        """Run this context's work on torch's current stream, so that torch ops, collectives and the library's
        kernels are ordered by the stream alone.  torch's default stream is the null stream, whose handle 0 means
        "back to the context's own stream" to ntt_ctx_set_stream: it is passed as hipStreamLegacy (handle 1)."""
        import torch
        s = torch.cuda.current_stream(self.device).cuda_stream
        self.set_stream(s if s else 1)

    @property
    def stream(self):
        return _lib.ntt_ctx_stream(self._h)

    def synchronize(self):
        _check(_lib.ntt_ctx_synchronize(self._h))

    def plan(self, p, n, batch=1, kind=MONTGOMERY):
        return Plan(self, p, n, batch, kind)

    def negacyclic_plan(self, p, n, batch=1, kind=MONTGOMERY):
        return NegacyclicPlan(self, p, n, batch, kind)

    def plan64(self, p64, n, batch=1):
        return Plan64(self, p64, n, batch)

    def negacyclic_plan64(self, p64, n, batch=1):
        return Plan64(self, p64, n, batch, negacyclic=True)

    def crt_multiply(self, a, b, p64, out=None):
        """Large-modulus product (main-openmp.cc:693-711): a, b uint64 < p64; returns 2n-1 uint64."""
        n = int(a.shape[-1]) if hasattr(a, "shape") else len(a)
        if out is None:
            out = np.zeros(2 * n - 1, dtype=np.uint64)
        _check(_lib.ntt_crt_multiply(self._h, _ptr(a, np.uint64, n), _ptr(b, np.uint64, n),
                                     _ptr(out, np.uint64, 2 * n - 1), n, int(p64)))
        return out

    def crt_multiply_batched(self, a, b, p64, out=None):
        """[batch, n] uint64 pairs -> [batch, 2n-1] uint64 (``ntt_crt_multiply_batched``)."""
        batch, n = int(a.shape[0]), int(a.shape[1])
        if out is None:
            out = np.zeros((batch, 2 * n - 1), dtype=np.uint64)
        _check(_lib.ntt_crt_multiply_batched(self._h, _ptr(a, np.uint64, batch * n), _ptr(b, np.uint64, batch * n),
                                             _ptr(out, np.uint64, batch * (2 * n - 1)), n, int(p64), batch))
        return out

    def crt_residue_multiply(self, a, b, prime_index, out=None):
        """Residue product mod the k-th CRT prime (CRT_parallel, main-openmp.cc:424-440): uint32[2n-1]."""
        n = int(a.shape[-1])
        if out is None:
            out = np.zeros(2 * n - 1, dtype=np.uint32)
        _check(_lib.ntt_crt_residue_multiply(self._h, _ptr(a, np.uint64, n), _ptr(b, np.uint64, n),
                                             _ptr(out, np.uint32, 2 * n - 1), n, int(prime_index)))
        return out

    def crt_merge(self, rows, p64, out=None, count=None):
        """CRT_combine_parallel (main-openmp.cc:468-485) on any slice: rows = four uint32 arrays."""
        if count is None:
            count = int(rows[0].shape[-1])
        if out is None:
            out = np.zeros(count, dtype=np.uint64)
        _check(_lib.ntt_crt_merge(self._h, *[_ptr(r, np.uint32, count) for r in rows], _ptr(out, np.uint64, count),
                                  count, int(p64)))
        return out

    def enable_kernel_timing(self, enable=True):
        _check(_lib.ntt_ctx_enable_kernel_timing(self._h, 1 if enable else 0))

    def kernel_timing_select(self, name=None):
        """Bracket only the kernel called `name` with events (None: every kernel again)."""
        index = -1
        if name is not None:
            i = 0
            while True:
                nm = C.c_char_p()
                if _lib.ntt_ctx_kernel_time(self._h, i, C.byref(nm), None, None) != OK:
                    raise ValueError(f"no kernel named {name}")
                if nm.value.decode() == name:
                    index = i
                    break
                i += 1
        _check(_lib.ntt_ctx_kernel_timing_select(self._h, index))

    def kernel_timing_stride(self, every=1):
        """Bracket only every `every`-th launch of the selected kernel(s)."""
        _check(_lib.ntt_ctx_kernel_timing_stride(self._h, int(every)))

    def kernel_times(self):
        """{kernel name: (total ms, launches)} accumulated since timing was (re-)enabled."""
        out = {}
        i = 0
        while True:
            name, ms, cnt = C.c_char_p(), C.c_double(), C.c_long()
            rc = _lib.ntt_ctx_kernel_time(self._h, i, C.byref(name), C.byref(ms), C.byref(cnt))
            if rc != OK:
                break
            out[name.value.decode()] = (ms.value, cnt.value)
            i += 1
        return out


class Plan:
    """Products of two length-n polynomials mod the prime p (``ntt_plan``)."""

    def __init__(self, ctx, p, n, batch=1, kind=MONTGOMERY):
        h = C.c_void_p()
        _check(_lib.ntt_plan_create(ctx._h, int(p), int(n), int(batch), int(kind), C.byref(h)))
        self._h = h
        self.ctx, self.p, self.n, self.batch, self.kind = ctx, int(p), int(n), int(batch), kind
        self.length = _lib.ntt_plan_length(h)
        ctx._plans.add(self)

    def close(self):
        if self._h:
            if self.ctx._h:                  # the context closes its plans first; never destroy against a dead one
                _lib.ntt_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def multiply(self, a, b, out=None, batch=None, out_stride=None):
        """a, b: [batch, n] uint32 -> [batch, out_stride] uint32 holding 2n-1 coefficients per row."""
        if batch is None:
            batch = int(a.shape[0]) if getattr(a, "ndim", 1) == 2 or (hasattr(a, "dim") and a.dim() == 2) else 1
        if out_stride is None:
            out_stride = 2 * self.n - 1
        if out is None:
            out = np.zeros((batch, out_stride), dtype=np.uint32)
        need_out = (batch - 1) * out_stride + 2 * self.n - 1
        _check(_lib.ntt_poly_multiply_batched(self._h, _ptr(a, np.uint32, batch * self.n), _ptr(b, np.uint32, batch * self.n),
                                              _ptr(out, np.uint32, need_out), out_stride, batch))
        return out

    def multiply_slots(self, a_list, b_list, out_list, counts, out_stride=None):
        """Plan of a multi-device context: slot s multiplies its own arrays a_list[s], b_list[s] ([counts[s], n]) into
        out_list[s] (``ntt_poly_multiply_batched_slots``); None for a slot without work."""
        nd = len(counts)
        if out_stride is None:
            out_stride = 2 * self.n - 1

        def arr(xs, need_of):
            return (C.c_void_p * nd)(*[_ptr(x, np.uint32, need_of(c)).value if c else None for x, c in zip(xs, counts)])
        ca = (C.c_int * nd)(*[int(c) for c in counts])
        _check(_lib.ntt_poly_multiply_batched_slots(
            self._h, arr(a_list, lambda c: c * self.n), arr(b_list, lambda c: c * self.n),
            arr(out_list, lambda c: (c - 1) * out_stride + 2 * self.n - 1), out_stride, ca))
        return out_list

    def multiply_noalloc(self, a, b, out, batch, out_stride=None):
        """``ntt_poly_multiply_batched_noalloc``: device tensors only, intermediates in the library's static
        scratch array, nothing allocated by the call; `batch` need not fit the plan's capacity."""
        if out_stride is None:
            out_stride = 2 * self.n - 1
        need_out = (batch - 1) * out_stride + 2 * self.n - 1
        _check(_lib.ntt_poly_multiply_batched_noalloc(self._h, _ptr(a, np.uint32, batch * self.n), _ptr(b, np.uint32, batch * self.n),
                                                      _ptr(out, np.uint32, need_out), out_stride, batch))
        return out

    def multiply_stream(self, a, b, out=None, out_stride=None):
        """Host arrays a, b: [count, n] uint32, any count; slices stream through the GPU (pinned arrays
        from ``host_array`` are DMA'd directly, pageable ones are staged).  Returns [count, out_stride]."""
        count = int(a.shape[0])
        if out_stride is None:
            out_stride = 2 * self.n - 1
        if out is None:
            out = np.zeros((count, out_stride), dtype=np.uint32)
        need_out = (count - 1) * out_stride + 2 * self.n - 1 if count else 0
        _check(_lib.ntt_poly_multiply_stream(self._h, _ptr(a, np.uint32, count * self.n), _ptr(b, np.uint32, count * self.n),
                                             _ptr(out, np.uint32, need_out), out_stride, count))
        return out

    def forward(self, data, batch=1):
        _check(_lib.ntt_forward(self._h, _ptr(data, np.uint32, batch * self.length), batch))
        return data

    def inverse(self, data, batch=1):
        _check(_lib.ntt_inverse(self._h, _ptr(data, np.uint32, batch * self.length), batch))
        return data


class NegacyclicPlan(Plan):
    """Products modulo (x^n + 1, p): ``ntt_negacyclic_plan_create`` / ``ntt_negacyclic_multiply_batched``."""

    def __init__(self, ctx, p, n, batch=1, kind=MONTGOMERY):
        h = C.c_void_p()
        _check(_lib.ntt_negacyclic_plan_create(ctx._h, int(p), int(n), int(batch), int(kind), C.byref(h)))
        self._h = h
        self.ctx, self.p, self.n, self.batch, self.kind = ctx, int(p), int(n), int(batch), kind
        self.length = _lib.ntt_plan_length(h)
        ctx._plans.add(self)

    def multiply(self, a, b, out=None, batch=None, out_stride=None):
        """a, b: [batch, n] uint32 -> [batch, out_stride] uint32 holding the n coefficients of a*b mod x^n+1."""
        if batch is None:
            batch = int(a.shape[0]) if getattr(a, "ndim", 1) == 2 or (hasattr(a, "dim") and a.dim() == 2) else 1
        if out_stride is None:
            out_stride = self.n
        if out is None:
            out = np.zeros((batch, out_stride), dtype=np.uint32)
        need_out = (batch - 1) * out_stride + self.n
        _check(_lib.ntt_negacyclic_multiply_batched(self._h, _ptr(a, np.uint32, batch * self.n), _ptr(b, np.uint32, batch * self.n),
                                                    _ptr(out, np.uint32, need_out), out_stride, batch))
        return out


class Plan64:
    """Products modulo one NTT-friendly prime below 2^62 with 64-bit lanes (``ntt_plan64``)."""

    def __init__(self, ctx, p64, n, batch=1, negacyclic=False):
        h = C.c_void_p()
        create = _lib.ntt_negacyclic_plan64_create if negacyclic else _lib.ntt_plan64_create
        _check(create(ctx._h, int(p64), int(n), int(batch), C.byref(h)))
        self._h = h
        self.ctx, self.p, self.n, self.batch, self.negacyclic = ctx, int(p64), int(n), int(batch), negacyclic
        L = C.c_uint32()
        _check(_lib.ntt_plan64_length(h, C.byref(L)))
        self.length = L.value
        ctx._plans.add(self)

    def close(self):
        if self._h:
            if self.ctx._h:
                _lib.ntt_plan64_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def multiply(self, a, b, out=None, batch=None, out_stride=None):
        """a, b: [batch, n] uint64 -> [batch, out_stride] uint64 holding 2n-1 coefficients per row."""
        if batch is None:
            batch = int(a.shape[0]) if getattr(a, "ndim", 1) == 2 or (hasattr(a, "dim") and a.dim() == 2) else 1
        if out_stride is None:
            out_stride = self.n if self.negacyclic else 2 * self.n - 1
        if out is None:
            out = np.zeros((batch, out_stride), dtype=np.uint64)
        fn = _lib.ntt_negacyclic_multiply64_batched if self.negacyclic else _lib.ntt_poly_multiply64_batched
        need_out = (batch - 1) * out_stride + (self.n if self.negacyclic else 2 * self.n - 1)
        _check(fn(self._h, _ptr(a, np.uint64, batch * self.n), _ptr(b, np.uint64, batch * self.n),
                  _ptr(out, np.uint64, need_out), out_stride, batch))
        return out


class DistPlan:
    """This rank's pieces of one length-2^lgL transform split over `world` GPUs (``ntt_dist_plan``)."""

    def __init__(self, ctx, p, lgL, world, rank, kind=MONTGOMERY):
        h = C.c_void_p()
        _check(_lib.ntt_dist_plan_create(ctx._h, int(p), int(lgL), int(world), int(rank), int(kind), C.byref(h)))
        self._h = h
        self.ctx, self.p, self.lgL, self.world, self.rank = ctx, int(p), int(lgL), int(world), int(rank)
        self.local_length = _lib.ntt_dist_local_length(h)
        ctx._plans.add(self)

    def close(self):
        if self._h:
            if self.ctx._h:
                _lib.ntt_dist_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def colpass(self, data, inverse=False):
        """In place on the [world][S/world] device array held after the first all-to-all."""
        _check(_lib.ntt_dist_colpass(self._h, _ptr(data, np.uint32, self.local_length), 1 if inverse else 0))
        return data

    def local(self, row, inverse=False):
        """S-point transform (natural order) of this rank's row, in place."""
        _check(_lib.ntt_dist_local(self._h, _ptr(row, np.uint32, self.local_length), 1 if inverse else 0))
        return row

    @property
    def transport(self):
        """(name, column chunks) of the in-library exchange of a multi-device plan; (None, 0) for one rank's plan."""
        t, c = C.c_int(), C.c_int()
        _check(_lib.ntt_dist_plan_transport(self._h, C.byref(t), C.byref(c)))
        return ({0: "p2p", 1: "copyin", 2: "copy", 3: "rccl"}.get(t.value), c.value)

    def _blocks(self, arrays):
        if len(arrays) != self.world:
            raise ValueError(f"expected {self.world} blocks, one per slot")
        return (C.c_void_p * self.world)(*[_ptr(x, np.uint32, self.local_length).value for x in arrays])

    def forward(self, blocks_in, blocks_out):
        """Multi-device plan: blocks_in[r] = x[r*S:(r+1)*S] on slot r -> blocks_out[r][k2] = X[r + world*k2] (queued)."""
        _check(_lib.ntt_dist_forward(self._h, self._blocks(blocks_in), self._blocks(blocks_out)))
        return blocks_out

    def inverse(self, blocks_in, blocks_out):
        _check(_lib.ntt_dist_inverse(self._h, self._blocks(blocks_in), self._blocks(blocks_out)))
        return blocks_out

    def multiply(self, a_blocks, b_blocks, ab_blocks):
        """Multi-device plan: the product of two polynomials of L/2 coefficients held as world/2 blocks each (block r on
        slot r) into world blocks of the L-1 product coefficients (``ntt_dist_multiply``; queued)."""
        half = self.world // 2
        if len(a_blocks) != half or len(b_blocks) != half:
            raise ValueError(f"expected {half} operand blocks (the upper half of the transform is zero padding)")
        pad = [None] * (self.world - half)
        arr = lambda xs: (C.c_void_p * self.world)(*[_ptr(x, np.uint32, self.local_length).value if x is not None else None for x in xs])
        _check(_lib.ntt_dist_multiply(self._h, arr(list(a_blocks) + pad), arr(list(b_blocks) + pad), self._blocks(ab_blocks)))
        return ab_blocks


class Comm:
    """One rank's communicator of a process-per-GPU job (``ntt_comm``, include/ntt_amd.h): the exchanges of the split
    transform and of the prime-per-rank large-modulus product inside the library, for a host program shaped like the
    reference's MPI driver (main-mpi.cc:645-797).  Every rank makes the same calls in the same order.

    ``Comm.rccl(ctx, id, world, rank)``: RCCL on device arrays (``Comm.rccl_id()`` on one rank, the bytes handed to
    the others by the host program); ``Comm.host(ctx, alltoall, world, rank)``: ``alltoall(send, recv)`` is the host
    program's own all-to-all over two uint8 numpy views of pinned host memory shaped [world, bytes_per_rank] (row h of
    send goes to rank h, row s of recv comes from rank s) -- MPI_Alltoall, or torch.distributed over gloo."""

    def __init__(self, ctx, handle, keep=None):
        self.ctx, self._h, self._keep = ctx, handle, keep
        w, r, k = C.c_int(), C.c_int(), C.c_int()
        _check(_lib.ntt_comm_size(handle, C.byref(w), C.byref(r), C.byref(k)))
        self.world, self.rank, self.kind = w.value, r.value, ("rccl" if k.value else "host")
        ctx._plans.add(self)                 # closed with (before) its context

    @staticmethod
    def rccl_id():
        buf = C.create_string_buffer(COMM_ID_BYTES)
        _check(_lib.ntt_comm_rccl_id(buf))
        return buf.raw

    @classmethod
    def rccl(cls, ctx, comm_id, world, rank):
        if len(comm_id) != COMM_ID_BYTES:
            raise ValueError(f"the RCCL id has {COMM_ID_BYTES} bytes")
        h = C.c_void_p()
        _check(_lib.ntt_comm_create_rccl(ctx._h, C.c_char_p(bytes(comm_id)), int(world), int(rank), C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def host(cls, ctx, alltoall, world, rank):
        world = int(world)

        def thunk(_user, send, recv, nbytes):
            try:
                shape = (world, int(nbytes))
                s = np.ctypeslib.as_array(C.cast(send, C.POINTER(C.c_uint8)), shape=shape)
                r = np.ctypeslib.as_array(C.cast(recv, C.POINTER(C.c_uint8)), shape=shape)
                alltoall(s, r)
                return 0
            except Exception:                # an exception must not unwind through the C frames above
                import traceback
                traceback.print_exc()
                return 1
        cb = ALLTOALL_FN(thunk)
        h = C.c_void_p()
        _check(_lib.ntt_comm_create_host(ctx._h, cb, None, world, int(rank), C.byref(h)))
        return cls(ctx, h, keep=cb)

    def close(self):
        if self._h:
            if self.ctx._h:
                _lib.ntt_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def dist_forward(self, dplan, x_local, out):
        """x_local = x[rank*S:(rank+1)*S] (device) -> out[k2] = X[rank + world*k2] (``ntt_comm_dist_forward``)."""
        S = dplan.local_length
        _check(_lib.ntt_comm_dist_forward(self._h, dplan._h, _ptr(x_local, np.uint32, S), _ptr(out, np.uint32, S)))
        return out

    def dist_inverse(self, dplan, X_local, out):
        S = dplan.local_length
        _check(_lib.ntt_comm_dist_inverse(self._h, dplan._h, _ptr(X_local, np.uint32, S), _ptr(out, np.uint32, S)))
        return out

    def dist_multiply(self, dplan, a_block, b_block, ab_block):
        """The product of two polynomials of L/2 coefficients: ranks below world/2 pass their block of a and of b (the
        others None), every rank receives its block of the L-1 product coefficients (``ntt_comm_dist_multiply``)."""
        S = dplan.local_length
        opt = lambda x: _ptr(x, np.uint32, S) if x is not None else None
        _check(_lib.ntt_comm_dist_multiply(self._h, dplan._h, opt(a_block), opt(b_block), _ptr(ab_block, np.uint32, S)))
        return ab_block

    def crt_multiply(self, a, b, p64, out=None):
        """Prime-per-rank large-modulus product; a, b (uint64 [n], the same on every rank) -> 2n-1 uint64 on every rank."""
        n = int(a.shape[-1])
        if out is None:
            out = np.zeros(2 * n - 1, dtype=np.uint64)
        _check(_lib.ntt_comm_crt_multiply(self._h, _ptr(a, np.uint64, n), _ptr(b, np.uint64, n),
                                          _ptr(out, np.uint64, 2 * n - 1), n, int(p64)))
        return out


class _PinnedOwner:
    def __init__(self, ptr):
        self.ptr = ptr

    def __del__(self):
        try:
            _lib.ntt_host_free(self.ptr)
        except Exception:
            pass


def host_array(shape, dtype=np.uint32):
    """numpy array over page-locked host memory (``ntt_host_alloc``); freed with its last view."""
    shape = tuple(int(x) for x in (shape if hasattr(shape, "__len__") else (shape,)))
    nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
    p = C.c_void_p()
    _check(_lib.ntt_host_alloc(max(nbytes, 1), C.byref(p)))
    owner = _PinnedOwner(p)
    buf = (C.c_char * max(nbytes, 1)).from_address(p.value)
    buf._owner = owner                     # the ctypes buffer (and every numpy view of it) keeps the memory alive
    return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)


def pack_case(directory, case_id):
    _check(_lib.ntt_pack_case(str(directory).encode(), case_id))


def read_case_bin(directory, case_id, cap=1 << 22):
    """-> n, p, a, b with a, b uint32 (p < 2^32) or uint64 arrays, as stored."""
    a = np.zeros(cap, dtype=np.uint64)
    b = np.zeros(cap, dtype=np.uint64)
    n, p, w = C.c_int(), C.c_uint64(), C.c_int()
    _check(_lib.ntt_read_case_bin(str(directory).encode(), case_id, a.ctypes.data, b.ctypes.data, cap * 8,
                                  C.byref(n), C.byref(p), C.byref(w)))
    dt = np.uint32 if w.value == 4 else np.uint64
    return n.value, p.value, a.view(dt)[:n.value].copy(), b.view(dt)[:n.value].copy()


def check_case_bin(directory, case_id, ab):
    ab = np.ascontiguousarray(ab)
    assert ab.dtype in (np.uint32, np.uint64)
    n = (ab.size + 1) // 2
    rc = _lib.ntt_check_case_bin(str(directory).encode(), case_id, ab.ctypes.data, n, ab.dtype.itemsize)
    if rc == ERR_MISMATCH:
        return False
    _check(rc)
    return True


def write_case_bin(directory, case_id, ab, p):
    ab = np.ascontiguousarray(ab)
    assert ab.dtype in (np.uint32, np.uint64)
    n = (ab.size + 1) // 2
    _check(_lib.ntt_write_case_bin(str(directory).encode(), case_id, ab.ctypes.data, n, int(p), ab.dtype.itemsize))


def multiply_limbs(plans, a_list, b_list, out_list, batch, out_stride=None):
    """Products under several moduli in one call (``ntt_poly_multiply_limbs``): limb i multiplies `batch` rows of
    a_list[i], b_list[i] into out_list[i] with plans[i] -- distinct plans of one context, device arrays; queued."""
    k = len(plans)
    if not (len(a_list) == len(b_list) == len(out_list) == k):
        raise ValueError("one array triple per plan")
    n = plans[0].n
    row = n if isinstance(plans[0], NegacyclicPlan) else 2 * n - 1
    if out_stride is None:
        out_stride = row
    arr = lambda xs, need: (C.c_void_p * k)(*[_ptr(x, np.uint32, need).value for x in xs])
    ph = (C.c_void_p * k)(*[pl._h.value for pl in plans])
    _check(_lib.ntt_poly_multiply_limbs(ph, k, arr(a_list, batch * n), arr(b_list, batch * n),
                                        arr(out_list, (batch - 1) * out_stride + row), out_stride, int(batch)))
    return out_list


def partition(total_units, world, rank):
    """(start, count) of slot `rank`'s units: the library's own sharding rule (``ntt_partition``, main-mpi.cc:657-660)."""
    st, cnt = C.c_long(), C.c_long()
    _check(_lib.ntt_partition(int(total_units), int(world), int(rank), C.byref(st), C.byref(cnt)))
    return st.value, cnt.value


def crt_prime(index):
    p = C.c_uint32()
    _check(_lib.ntt_crt_prime(index, C.byref(p)))
    return p.value


def poly_multiply(a, b, p):
    """The reference entry point: int32 arrays a, b of length n -> int32 array of 2n-1 products mod p."""
    a = np.ascontiguousarray(a, dtype=np.int32)
    b = np.ascontiguousarray(b, dtype=np.int32)
    n = a.size
    ab = np.zeros(2 * n - 1, dtype=np.int32)
    _lib.poly_multiply(a.ctypes.data, b.ctypes.data, ab.ctypes.data, n, int(p))
    return ab


def read_case(directory, case_id, cap=1 << 22):
    a = np.zeros(cap, dtype=np.uint64)
    b = np.zeros(cap, dtype=np.uint64)
    n, p = C.c_int(), C.c_uint64()
    _check(_lib.ntt_read_case(str(directory).encode(), case_id, a.ctypes.data, b.ctypes.data, cap,
                              C.byref(n), C.byref(p)))
    return n.value, p.value, a[:n.value].copy(), b[:n.value].copy()


def check_case(directory, case_id, ab):
    """Prints the reference's verdict line; returns True / False; raises on I/O errors."""
    ab = np.ascontiguousarray(ab, dtype=np.uint64)
    n = (ab.size + 1) // 2
    rc = _lib.ntt_check_case(str(directory).encode(), case_id, ab.ctypes.data, n)
    if rc == ERR_MISMATCH:
        return False
    _check(rc)
    return True


def write_case(directory, case_id, ab):
    ab = np.ascontiguousarray(ab, dtype=np.uint64)
    n = (ab.size + 1) // 2
    _check(_lib.ntt_write_case(str(directory).encode(), case_id, ab.ctypes.data, n))
