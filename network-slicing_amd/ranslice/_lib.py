"""ctypes binding of libranslice.so (include/ranslice.h).  Fails loudly when the HIP library is
missing or no GPU is usable: there is NO CPU fallback in the product path."""
import ctypes as C
import os

from .config import RsConfig, KbConfig

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('RANSLICE_LIB', os.path.join(os.path.dirname(_HERE), 'csrc', 'build', 'libranslice.so'))
# the test build (make dev: -DRS_DEV): the same sources with the developer knobs -- sweep switches, guard bands, the fault injector
# of the shared step -- readable from the environment.  Tests that turn a knob ask for it (load(dev=True), or RANSLICE_DEV_BUILD=1
# in the environment when the handle is created); the production library reads no such variable.
DEV_LIB_PATH = os.path.join(os.path.dirname(LIB_PATH), 'libranslice_dev.so')

RS_OK, RS_EINVAL, RS_EOVERFLOW, RS_EHIP, RS_ESTATE = 0, -1, -2, -3, -4

KB_EXPORTS = (
    'kb_create', 'kb_destroy', 'kb_last_error', 'kb_reset', 'kb_update_control', 'kb_select_action',
    'kb_step_resident', 'kb_run_resident', 'kb_predict', 'kb_update', 'kb_get_learner', 'kb_get_control', 'kb_set_adjusted',
    'kb_comm_info', 'kb_get_stats', 'kb_get_sizes', 'kb_get_pool', 'kb_state_bytes', 'kb_save_state', 'kb_load_state', 'kb_get_flags', 'kb_get_repair_work', 'kb_get_kernel_row', 'kb_shared_scan', 'kb_shared_apply', 'kb_shared_commit', 'kb_comm_unique_id', 'kb_comm_init', 'kb_shared_step', 'kb_shared_step_resident', 'kb_shared_merge', 'kb_history_begin', 'kb_history_fetch', 'kb_kernel_time_ms', 'kb_phase_times_ms', 'kb_repair_times_ms', 'kb_kernel_times_ms', 'kb_set_kernel_timing', 'kb_synchronize',
)

EXPORTS = (
    'rs_create', 'rs_load_fading', 'rs_reset', 'rs_step', 'rs_step_resident', 'rs_random_actions', 'rs_fetch',
    'rs_get_info', 'rs_set_alloc_trace', 'rs_get_alloc_trace', 'rs_get_counters', 'rs_get_rx_stats', 'rs_set_group_size', 'rs_set_schedule_hint', 'rs_get_section_profile', 'rs_get_task_profile', 'rs_run_random', 'rs_kernel_time_ms', 'rs_kernel_time_stats_ms',
    'rs_set_kernel_timing', 'rs_synchronize', 'rs_state_bytes', 'rs_save_state', 'rs_load_state', 'rs_device_count', 'rs_device_mem_info', 'rs_n_vars', 'rs_n_slices', 'rs_last_error', 'rs_destroy',
) + KB_EXPORTS


class RanSliceError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__('libranslice error %d: %s' % (code, msg))
        self.code = code


_libs = {}


def device_count():
    """HIP devices visible to this process"""
    n = load().rs_device_count()
    if n < 0:
        raise RanSliceError(n, 'hipGetDeviceCount failed')
    return n


def device_mem_info(device=0):
    """(free, total) bytes of the device's memory"""
    f, t = C.c_uint64(), C.c_uint64()
    rc = load().rs_device_mem_info(int(device), C.byref(f), C.byref(t))
    if rc < 0:
        raise RanSliceError(rc, 'hipMemGetInfo failed')
    return int(f.value), int(t.value)


def default_pool_bytes(device=0, headroom=24 << 30, floor=8 << 30):
    """a KBRL pool that takes what the device has left after `headroom` (the simulator's state, the tables, the histories)"""
    free, _ = device_mem_info(device)
    return max(int(floor), int(free) - int(headroom))


def load(dev=None):
    """Load libranslice.so; raises if it has not been built (python __graft_entry__.py build).  dev=True (or
    RANSLICE_DEV_BUILD=1 in the environment at the time of the call) loads the test build instead; all handles that meet in
    one call (an agent stepping an environment) must come from the same one."""
    if dev is None:
        dev = os.environ.get('RANSLICE_DEV_BUILD') == '1'
    path = DEV_LIB_PATH if dev else LIB_PATH
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise ImportError('%s not found: build it with `make -C network-slicing_amd/csrc` '
                          '(hipcc, gfx950). There is no CPU fallback.' % path)
    # A handle drives two or three HIP streams (the simulator's, its mMTC side stream, the agent's) and several handles may run
    # side by side in one process (experiments_kbrl.evaluate_grid: six cells = 18 streams).  The HIP runtime multiplexes
    # streams onto 4 hardware queues unless told otherwise, and streams that share a queue run one after the other: the
    # six-cell grid took 26.5 s per 6,000 steps on 4 queues, 16.3 s on 24 (profiles/r04_g_queues.txt).  Read by the runtime
    # at its first call, so it has to be in the environment before the library initialises; a value the user set stays.
    os.environ.setdefault('GPU_MAX_HW_QUEUES', '24')
    L = C.CDLL(path)
    vp, ip, dp = C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_double)
    fp, up = C.POINTER(C.c_float), C.POINTER(C.c_uint64)
    L.rs_create.argtypes = [C.POINTER(RsConfig), C.c_int, C.POINTER(vp)]
    L.rs_load_fading.argtypes = [vp, C.c_int, dp, C.c_int, C.c_int]
    L.rs_reset.argtypes = [vp, up, fp]
    L.rs_step.argtypes = [vp, ip, fp, dp, ip, ip]
    L.rs_step_resident.argtypes = [vp]
    L.rs_random_actions.argtypes = [vp, C.c_uint64, C.c_uint64]
    L.rs_run_random.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_int, C.c_int]
    L.rs_fetch.argtypes = [vp, ip, fp, dp, ip, ip]
    L.rs_get_info.argtypes = [vp, dp]
    L.rs_set_alloc_trace.argtypes = [vp, C.c_int]
    L.rs_get_alloc_trace.argtypes = [vp, vp]
    L.rs_get_counters.argtypes = [vp, up]
    L.rs_get_rx_stats.argtypes = [vp, up]
    L.rs_get_section_profile.argtypes = [vp, up]
    L.rs_get_task_profile.argtypes = [vp, up]
    L.rs_set_group_size.argtypes = [vp, C.c_int]
    L.rs_set_schedule_hint.argtypes = [vp, C.c_int]
    L.rs_kernel_time_ms.argtypes = [vp, dp, C.POINTER(C.c_int64)]
    L.rs_kernel_time_stats_ms.argtypes = [vp, dp, C.POINTER(C.c_int64)]
    L.rs_set_kernel_timing.argtypes = [vp, C.c_int]
    L.rs_synchronize.argtypes = [vp]
    L.rs_n_vars.argtypes = [vp]
    L.rs_n_slices.argtypes = [vp]
    L.rs_last_error.argtypes = [vp]
    L.rs_last_error.restype = C.c_char_p
    L.rs_destroy.argtypes = [vp]
    L.rs_destroy.restype = None
    i64p = C.POINTER(C.c_int64)
    L.kb_create.argtypes = [C.POINTER(KbConfig), C.c_int, C.POINTER(vp)]
    L.kb_destroy.argtypes = [vp]
    L.kb_destroy.restype = None
    L.kb_last_error.argtypes = [vp]
    L.kb_last_error.restype = C.c_char_p
    L.kb_reset.argtypes = [vp, ip, ip, up]
    L.kb_update_control.argtypes = [vp, fp, ip, ip, ip]
    L.kb_select_action.argtypes = [vp, fp, ip, ip]
    L.kb_step_resident.argtypes = [vp, vp]
    L.kb_run_resident.argtypes = [vp, vp, C.c_int, C.c_int]
    L.kb_predict.argtypes = [vp, C.c_int, C.c_int, dp, ip, dp]
    L.kb_update.argtypes = [vp, C.c_int, C.c_int, dp, C.c_int32, ip, dp]
    L.kb_get_learner.argtypes = [vp, C.c_int, C.c_int, ip, dp, dp, dp]
    L.kb_get_control.argtypes = [vp, ip, ip, ip, ip, dp]
    L.kb_set_adjusted.argtypes = [vp, ip]
    L.kb_get_stats.argtypes = [vp, up]
    L.kb_get_sizes.argtypes = [vp, ip]
    L.kb_get_pool.argtypes = [vp, up, up, ip, ip]
    L.kb_get_repair_work.argtypes = [vp, up]
    for _n in ('rs', 'kb'):
        getattr(L, _n + '_state_bytes').argtypes = [vp, up]
        getattr(L, _n + '_save_state').argtypes = [vp, vp, C.c_uint64]
        getattr(L, _n + '_load_state').argtypes = [vp, vp, C.c_uint64]
    L.kb_get_flags.argtypes = [vp, ip]
    L.kb_get_kernel_row.argtypes = [vp, C.c_int, C.c_int, ip, dp]
    L.kb_shared_scan.argtypes = [vp, fp, ip, ip, C.c_int32, C.c_int32, ip, ip, dp]
    L.kb_shared_apply.argtypes = [vp, ip, dp, C.c_int32]
    L.kb_shared_commit.argtypes = [vp, ip]
    sp = C.POINTER(C.c_int16)
    L.kb_history_begin.argtypes = [vp, C.c_int32]
    L.kb_history_fetch.argtypes = [vp, dp, sp, sp, sp, sp, sp, ip]
    L.kb_comm_unique_id.argtypes = [vp]
    L.kb_comm_init.argtypes = [vp, vp, C.c_int, C.c_int]
    L.kb_comm_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.kb_shared_step.argtypes = [vp, fp, ip, ip, C.c_int32, C.c_int32, ip, ip]
    L.kb_shared_step_resident.argtypes = [vp, vp, C.c_int32, C.c_int32, ip]
    L.kb_shared_merge.argtypes = [vp, dp, C.c_int32, C.c_int32, C.c_int32, dp, ip, ip, ip]
    L.rs_device_count.argtypes = []
    L.rs_device_mem_info.argtypes = [C.c_int, up, up]
    L.kb_kernel_time_ms.argtypes = [vp, dp, i64p]
    L.kb_phase_times_ms.argtypes = [vp, dp, i64p]
    L.kb_repair_times_ms.argtypes = [vp, dp, i64p]
    L.kb_kernel_times_ms.argtypes = [vp, dp, i64p]
    L.kb_set_kernel_timing.argtypes = [vp, C.c_int]
    L.kb_synchronize.argtypes = [vp]
    for name in EXPORTS:
        if name not in ('rs_last_error', 'rs_destroy', 'kb_last_error', 'kb_destroy'):
            getattr(L, name).restype = C.c_int
    _libs[path] = L
    return L
