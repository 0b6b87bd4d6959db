"""ctypes binding of libmetalens_hip.so (include/metalens_hip.h).

There is no CPU fallback: if the shared library is missing or no MI355X is
visible, every entry point of the package raises ``MetalensHipError``.
"""
import ctypes
import os
from ctypes import (POINTER, Structure, byref, c_char_p, c_double, c_int, c_int32, c_int64,
                    c_uint8, c_void_p)

import numpy as np

# The host driver of this GPU pool only supports dmabuf IPC; RCCL (multi-process runs) fails with
# "hipIpcGetMemHandle: invalid argument" unless this is set before the HIP runtime initialises.
# Harmless for single-process use; an explicit setting by the user wins.
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

LIB_NAME = 'libmetalens_hip.so'
# METALENS_HIP_LIB: another build of the same library (diagnostic builds, A/B timing)
LIB_PATH = os.environ.get('METALENS_HIP_LIB') or os.path.join(
    os.path.dirname(os.path.abspath(__file__)), LIB_NAME)

TIE_CAPACITY = 1048576   # ML_TIE_CAPACITY

K_NEARFIELD, K_TWIDDLE, K_ZGEMM_STAGE1, K_ZGEMM_STAGE2, K_PROJECT, K_LATTICE_POWER, K_COLDOT = range(7)
KERNEL_NAMES = ('nearfield', 'twiddle', 'zgemm_stage1', 'zgemm_stage2', 'project',
                'lattice_power', 'coldot', 'comm_wait', 'collective')

# every symbol include/metalens_hip.h declares (tests/test_cabi_symbols.py checks the header
# against this list and the built library)
SYMBOLS = (
    'ml_abi_version', 'ml_last_error', 'ml_device_count', 'ml_ctx_create', 'ml_ctx_destroy',
    'ml_device_info', 'ml_upload_table', 'ml_upload_layout', 'ml_nearfield',
    'ml_fields_download', 'ml_fields_upload', 'ml_fields_shape', 'ml_farfield_lattice_power',
    'ml_farfield_plan', 'ml_farfield_transform', 'ml_farfield_allreduce', 'ml_farfield_project',
    'ml_farfield_download', 'ml_farfield_plan_info', 'ml_farfield_set_precision', 'ml_farfield_project_reduce', 'ml_profile_select',
    'ml_profile_sample',
    'ml_nearfield_premodulate', 'ml_comm_unique_id', 'ml_comm_init', 'ml_comm_allreduce_host',
    'ml_comm_barrier', 'ml_profile_enable', 'ml_profile_reset', 'ml_profile_get', 'ml_sync',
    'ml_nearfield_async', 'ml_farfield_transform_async', 'ml_farfield_transform_mirrored',
    'ml_farfield_transform_mirrored_async', 'ml_farfield_project_async',
    'ml_nearfield_result', 'ml_nearfield_ties', 'ml_nearfield_tie_answers',
    'ml_farfield_plan_kernels', 'ml_farfield_set_method',
    'ml_farfield_interleave_block', 'ml_farfield_transform_interleaved_async',
    'ml_nearfield_batch_async', 'ml_fields_select', 'ml_nearfield_powers',
    'ml_farfield_accumulate', 'ml_farfield_sums', 'ml_farfield_total_power', 'ml_host_alloc', 'ml_host_free',
    'ml_comm_info', 'ml_comm_set_reduce', 'ml_farfield_gather', 'ml_nearfield_kernel_info',
    'ml_comm_set_max_channels',
)


class MetalensHipError(RuntimeError):
    pass


class NearfieldParams(Structure):
    _fields_ = [('source_x', c_double), ('source_y', c_double), ('source_z', c_double),
                ('dz', c_double), ('dz2', c_double), ('source_z2', c_double),
                ('pol', c_double * 3), ('kvac', c_double), ('kvac2', c_double),
                ('k_glass', c_double), ('k_glass2', c_double), ('n_glass', c_double),
                ('Z0', c_double), ('H_coef', c_double), ('dipole_moment', c_double),
                ('plane_wave', c_int32), ('reserved', c_int32)]


class BoundViolation(Structure):
    _fields_ = [('slot', c_int32), ('order', c_int32), ('check', c_int32),
                ('reserved', c_int32), ('value', c_double), ('bound', c_double)]


_lib = None
_dp = POINTER(c_double)
_ip = POINTER(c_int32)


def load():
    """dlopen the library once and declare the argument types."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MetalensHipError(
            '%s is not built; run `python -c "import __graft_entry__ as g; g.build()"` or '
            '`make -C metalens_amd/csrc` (needs hipcc, cross-compiles for gfx950 without a GPU)'
            % LIB_PATH)
    try:
        lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    except OSError as e:
        raise MetalensHipError('cannot load %s: %s' % (LIB_PATH, e))
    lib.ml_last_error.restype = c_char_p
    lib.ml_ctx_destroy.restype = None
    lib.ml_ctx_create.argtypes = [c_int, POINTER(c_void_p)]
    lib.ml_ctx_destroy.argtypes = [c_void_p]
    lib.ml_device_count.argtypes = [POINTER(c_int)]
    lib.ml_device_info.argtypes = [c_void_p, c_char_p, c_int, POINTER(c_int), POINTER(c_int64)]
    lib.ml_upload_table.argtypes = [c_void_p, c_int, _dp, c_int, _dp, c_int, _dp, c_int, _ip, _dp,
                                    c_int, _dp, _dp, _dp]
    lib.ml_upload_layout.argtypes = [c_void_p, c_int, _dp, _dp, _dp, _dp, _dp, _ip, _dp, _dp, c_int,
                                     _ip, _ip, c_int, _dp]
    lib.ml_nearfield.argtypes = [c_void_p, POINTER(NearfieldParams), _dp, c_int, _dp, c_int, _dp,
                                 POINTER(BoundViolation), c_int, POINTER(c_int)]
    lib.ml_nearfield_async.argtypes = [c_void_p, POINTER(NearfieldParams), _dp, c_int, _dp, c_int]
    lib.ml_nearfield_batch_async.argtypes = [c_void_p, POINTER(NearfieldParams), c_int, _dp, c_int,
                                             _dp, c_int]
    lib.ml_fields_select.argtypes = [c_void_p, c_int]
    lib.ml_nearfield_powers.argtypes = [c_void_p, _dp, c_int]
    lib.ml_farfield_accumulate.argtypes = [c_void_p, c_double, c_double, c_double, c_double, c_int, c_int]
    lib.ml_farfield_sums.argtypes = [c_void_p, _dp, _dp, _dp, c_int]
    lib.ml_farfield_total_power.argtypes = [c_void_p, _dp]
    lib.ml_host_alloc.argtypes = [ctypes.c_uint64, POINTER(c_void_p)]
    lib.ml_host_free.argtypes = [c_void_p]
    lib.ml_nearfield_result.argtypes = [c_void_p, _dp, POINTER(BoundViolation), c_int,
                                        POINTER(c_int)]
    lib.ml_nearfield_ties.argtypes = [c_void_p, POINTER(c_int64), c_int, POINTER(c_int)]
    lib.ml_nearfield_tie_answers.argtypes = [c_void_p, POINTER(c_int64), POINTER(c_int32), c_int]
    lib.ml_fields_download.argtypes = [c_void_p, _dp, _dp, _dp, _dp]
    lib.ml_fields_upload.argtypes = [c_void_p, c_int, c_int, _dp, _dp, _dp, _dp]
    lib.ml_fields_shape.argtypes = [c_void_p, POINTER(c_int), POINTER(c_int)]
    lib.ml_farfield_lattice_power.argtypes = [c_void_p, c_int, c_int, _dp, _dp, _dp, _dp, _dp, _dp,
                                              c_double, c_double, c_double, c_double, c_double, _dp]
    lib.ml_farfield_plan.argtypes = [c_void_p, c_int, c_int, c_double, c_double, c_double,
                                     c_double, _dp, c_int, _dp, c_int, c_int]
    lib.ml_farfield_transform.argtypes = [c_void_p, c_int, c_int]
    lib.ml_farfield_transform_async.argtypes = [c_void_p, c_int, c_int]
    lib.ml_farfield_transform_mirrored.argtypes = [c_void_p, c_int, c_int]
    lib.ml_farfield_transform_mirrored_async.argtypes = [c_void_p, c_int, c_int]
    lib.ml_farfield_allreduce.argtypes = [c_void_p]
    lib.ml_farfield_project.argtypes = [c_void_p, c_double, _dp, _dp, _dp]
    lib.ml_farfield_project_async.argtypes = [c_void_p, c_double]
    lib.ml_farfield_download.argtypes = [c_void_p, _dp, _dp, _dp, _dp]
    lib.ml_farfield_plan_info.argtypes = [c_void_p, POINTER(c_int)]
    lib.ml_farfield_set_precision.argtypes = [c_void_p, c_int]
    lib.ml_farfield_plan_kernels.argtypes = [c_void_p, POINTER(c_int), POINTER(c_int)]
    lib.ml_farfield_set_method.argtypes = [c_void_p, c_int]
    lib.ml_farfield_interleave_block.argtypes = [c_void_p, c_int, POINTER(c_int)]
    lib.ml_farfield_transform_interleaved_async.argtypes = [c_void_p, c_int, c_int, c_int, c_int]
    lib.ml_farfield_project_reduce.argtypes = [c_void_p, c_double]
    lib.ml_profile_select.argtypes = [c_void_p, ctypes.c_uint]
    lib.ml_profile_sample.argtypes = [c_void_p, c_int]
    lib.ml_nearfield_premodulate.argtypes = [c_void_p, c_int]
    lib.ml_comm_unique_id.argtypes = [POINTER(c_uint8)]
    lib.ml_comm_init.argtypes = [c_void_p, POINTER(c_uint8), c_int, c_int]
    lib.ml_comm_allreduce_host.argtypes = [c_void_p, _dp, c_int, c_int]
    lib.ml_comm_barrier.argtypes = [c_void_p]
    # (A/B timing against an older build of the library, METALENS_HIP_LIB: it may lack the newest entries)
    if hasattr(lib, 'ml_comm_info'):
        lib.ml_comm_info.argtypes = [c_void_p, POINTER(c_int), POINTER(c_int), POINTER(c_int)]
        lib.ml_comm_set_reduce.argtypes = [c_void_p, c_int]
        lib.ml_farfield_gather.argtypes = [c_void_p]
    if hasattr(lib, 'ml_comm_set_max_channels'):
        lib.ml_comm_set_max_channels.argtypes = [c_void_p, c_int]
    if hasattr(lib, 'ml_nearfield_kernel_info'):
        lib.ml_nearfield_kernel_info.argtypes = [c_void_p, POINTER(c_int), POINTER(c_int), POINTER(c_int)]
    lib.ml_profile_enable.argtypes = [c_void_p, c_int]
    lib.ml_profile_reset.argtypes = [c_void_p]
    lib.ml_profile_get.argtypes = [c_void_p, c_int, POINTER(c_int64), _dp]
    lib.ml_sync.argtypes = [c_void_p]
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise MetalensHipError('libmetalens_hip error %d: %s'
                               % (rc, load().ml_last_error().decode(errors='replace')))


def dptr(a):
    """pointer to a C-contiguous float64 / complex128 array (or NULL for None)"""
    if a is None:
        return None
    assert a.flags['C_CONTIGUOUS'] and a.dtype in (np.float64, np.complex128)
    return a.ctypes.data_as(_dp)


def iptr(a):
    assert a.flags['C_CONTIGUOUS'] and a.dtype == np.int32
    return a.ctypes.data_as(_ip)


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def c128(a):
    return np.ascontiguousarray(a, dtype=np.complex128)


class _PinnedPool:
    """Page-locked buffers behind the arrays the drop-in functions return.  Locking pages is slow
    (tens of ms for a few hundred MB), copying into locked pages is fast: a buffer goes back to
    the pool when the array built on it is garbage-collected, and the next call of the same size
    takes it from there."""

    def __init__(self, keep_bytes=None):
        self.free = {}          # nbytes -> [pointer, ...]
        self.kept = 0
        # page-locked memory kept for re-use (METALENS_PINNED_KEEP_BYTES overrides; 0 = keep none)
        if keep_bytes is None:
            keep_bytes = int(os.environ.get('METALENS_PINNED_KEEP_BYTES', 2 << 30))
        self.keep_bytes = keep_bytes

    def drain(self):
        """give every recycled page-locked buffer back to the system"""
        for stack in self.free.values():
            for ptr in stack:
                load().ml_host_free(c_void_p(ptr))
        self.free = {}
        self.kept = 0

    def _release(self, ptr, nbytes):
        if self.kept + nbytes <= self.keep_bytes:
            self.free.setdefault(nbytes, []).append(ptr)
            self.kept += nbytes
        else:
            try:
                load().ml_host_free(c_void_p(ptr))
            except Exception:   # interpreter shutdown
                pass

    def empty(self, shape, dtype):
        import weakref
        dtype = np.dtype(dtype)
        nbytes = int(np.prod(shape)) * dtype.itemsize
        if nbytes == 0:
            return np.empty(shape, dtype=dtype)
        stack = self.free.get(nbytes)
        if stack:
            ptr = stack.pop()
            self.kept -= nbytes
        else:
            p = c_void_p()
            if load().ml_host_alloc(nbytes, byref(p)) != 0 or not p.value:
                # the host cannot pin that much (more): first hand the recycled buffers back,
                # then fall back to pageable memory (slower copies, same results)
                self.drain()
                if load().ml_host_alloc(nbytes, byref(p)) != 0 or not p.value:
                    return np.empty(shape, dtype=dtype)
            ptr = p.value
        buf = (ctypes.c_char * nbytes).from_address(ptr)
        arr = np.frombuffer(buf, dtype=dtype).reshape(shape)
        # `buf` is the base object of every view of `arr`: it dies when the last of them does
        weakref.finalize(buf, self._release, ptr, nbytes)
        return arr


pinned = _PinnedPool()


class Context:
    """One GPU.  Owns the device-resident tables, layout, field set and far-field
    plan.  Not thread-safe; use one per host thread."""

    def __init__(self, device=None):
        lib = load()
        if device is None:
            device = int(os.environ.get('LOCAL_RANK', '0'))
        n = c_int(0)
        check(lib.ml_device_count(byref(n)))
        if n.value < 1:
            raise MetalensHipError('no HIP device visible: metalens_amd runs on MI355X (gfx950) '
                                   'only and has no CPU path')
        self._h = c_void_p()
        check(lib.ml_ctx_create(device % n.value, byref(self._h)))
        self.device = device % n.value
        self.lib = lib
        self.layout_token = None
        self.tables_token = None

    def close(self):
        if getattr(self, '_h', None):
            self.lib.ml_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        if not self._h:
            raise MetalensHipError('context is closed')
        return self._h

    def device_info(self):
        name = ctypes.create_string_buffer(64)
        cu = c_int(0)
        mem = c_int64(0)
        check(self.lib.ml_device_info(self.handle, name, 64, byref(cu), byref(mem)))
        return {'arch': name.value.decode(), 'cu_count': cu.value, 'hbm_bytes': mem.value}

    def sync(self):
        check(self.lib.ml_sync(self.handle))

    def set_precision(self, precision):
        """'f64' (default) or 'f32': arithmetic of the folded aperture -> direction GEMMs
        (include/metalens_hip.h, ml_farfield_set_precision); everything else stays fp64"""
        check(self.lib.ml_farfield_set_precision(self.handle, {'f64': 0, 'f32': 1}[precision]))
        self.precision = precision

    def set_method(self, method):
        """'auto' (default): axes whose direction grid sits on the aperture's FFT lattice run as
        output-pruned FFTs, the others as GEMMs; 'gemm': GEMMs everywhere; 'fft-streamed': as
        'auto' with the stage-1 result transposed for a streaming stage 2 at every size (auto:
        from 96 MiB of records + stage-1 result on).  Applies to the next plan."""
        check(self.lib.ml_farfield_set_method(self.handle, {'auto': 0, 'gemm': 1, 'fft-streamed': 2}[method]))

    def plan_kernels(self):
        """(stage 1, stage 2) of the active plan: 'gemm', 'folded' or 'fft'"""
        s1, s2 = c_int(0), c_int(0)
        check(self.lib.ml_farfield_plan_kernels(self.handle, byref(s1), byref(s2)))
        names = ('gemm', 'folded', 'fft')
        return names[s1.value], names[s2.value]

    def nearfield_kernels(self):
        """which synthesis kernels the last synthesis took: {'family': 'orders-along-x' (every table holds
        orders (ox, 0), |ox| <= 5: per-collection order lists, phasors by products), 'general', or 'mixed' (decided
        per table: the samples of simple tables as the former, of the others through the general kernel),
        'ring_orders_max', 'centre_orders'}"""
        fam, ring, cen = c_int(0), c_int(0), c_int(0)
        if not hasattr(self.lib, 'ml_nearfield_kernel_info'):   # (A/B runs against an older build)
            return {'family': 'unknown', 'ring_orders_max': None, 'centre_orders': None}
        check(self.lib.ml_nearfield_kernel_info(self.handle, byref(fam), byref(ring), byref(cen)))
        return {'family': ('general', 'orders-along-x', 'mixed')[fam.value], 'ring_orders_max': ring.value,
                'centre_orders': cen.value}

    def profile(self, on=True, kernels=None, every=1):
        """time kernel launches with HIP events; ``kernels`` = names to time (default all),
        ``every`` = time only every n-th launch of each (a sample, at 1/n of the cost)"""
        mask = 0xffffffff if kernels is None else sum(1 << KERNEL_NAMES.index(k) for k in kernels)
        check(self.lib.ml_profile_select(self.handle, mask))
        check(self.lib.ml_profile_sample(self.handle, int(every)))
        check(self.lib.ml_profile_enable(self.handle, int(on)))

    def profile_reset(self):
        check(self.lib.ml_profile_reset(self.handle))

    def profile_get(self):
        out = {}
        for k, name in enumerate(KERNEL_NAMES):
            n = c_int64(0)
            ms = c_double(0)
            check(self.lib.ml_profile_get(self.handle, k, byref(n), byref(ms)))
            out[name] = {'launches': n.value, 'total_ms': ms.value}
        return out


_default = None


def default_context():
    """process-wide context on device LOCAL_RANK (or 0)"""
    global _default
    if _default is None or not _default._h:
        _default = Context()
    return _default
