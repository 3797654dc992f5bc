"""VecRanSlice: N independent RanSlice environments advanced together on one MI355X.

This is the batched form of the reference's gym surface (reference
gym-ran_slice/gym_ran_slice/ran_slice.py:15-54): `reset()` and `step(actions)` with a leading
replica axis.  `gym_ran_slice.RanSlice` is its N=1 view.  All simulation runs in the HIP kernels
behind libranslice.so; this module only moves numpy buffers across the C ABI.
"""
import ctypes as C

import numpy as np

from . import _lib
from .config import RsAllocRec, make_config, n_vars
from .fading import synth_fading
from .sharding import replica_seeds

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_fp = C.POINTER(C.c_float)
_up = C.POINTER(C.c_uint64)


def default_fading(n_cols=10000, seed=20240):
    """The build's seeded stand-in for the three absent ns-3 traces."""
    return [synth_fading(t, n_cols, seed=seed) for t in range(3)]


class VecRanSlice:
    def __init__(self, n_envs=1, scenario=0, seed=0, device=0, fading=None, cfg=None, **cfg_kw):
        self.L = _lib.load()
        self.cfg = cfg if cfg is not None else make_config(scenario, n_envs=n_envs, **cfg_kw)
        self.cfg.n_envs = n_envs
        self.n_envs = n_envs
        self.n_ran = self.cfg.n_embb + self.cfg.n_mmtc
        # action / label entries per replica: one per L1 slice (with L1_level=False all eMBB RAN slices share one L1
        # slice and all mMTC ones another, scenario_creator.py:168-177)
        self.multiplexed = bool(self.cfg.l1_multiplex)
        self.n_slices = ((self.cfg.n_embb > 0) + (self.cfg.n_mmtc > 0)) if self.multiplexed else self.n_ran
        self.n_variables = n_vars(self.cfg)
        self.n_prbs = self.cfg.n_prbs
        self.penalty = self.cfg.penalty
        self.h = C.c_void_p()
        rc = self.L.rs_create(C.byref(self.cfg), int(device), C.byref(self.h))
        self._check(rc)
        if self.cfg.n_embb > 0:
            if fading is None:
                fading = default_fading()
            for t, tab in enumerate(fading):
                tab = np.ascontiguousarray(tab, dtype=np.float64)
                self._check(self.L.rs_load_fading(self.h, t, tab.ctypes.data_as(_dp), tab.shape[0], tab.shape[1]))
        self.base_seed = int(seed)
        self._obs = np.zeros((n_envs, self.n_variables), dtype=np.float32)
        self._reward = np.zeros(n_envs, dtype=np.float64)
        self._labels = np.zeros((n_envs, self.n_slices), dtype=np.int32)
        self._viol = np.zeros((n_envs, self.n_slices), dtype=np.int32)

    def _check(self, rc):
        if rc != 0:
            msg = self.L.rs_last_error(self.h).decode() if self.h else 'rs_create failed'
            raise _lib.RanSliceError(rc, msg)

    def close(self):
        if getattr(self, 'h', None):
            self.L.rs_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- gym-like surface -------------------------------------------------------------
    def reset(self, seeds=None):
        """seeds: per-replica 64-bit stream seeds; default sharding.replica_seeds(base_seed, 0, n_envs), a mix of
        the batch seed and the replica index (the reference seeds run i with default_rng(seed=i),
        experiments_kbrl.py:46)."""
        if seeds is None:
            seeds = replica_seeds(self.base_seed, 0, self.n_envs)
        seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
        assert seeds.shape == (self.n_envs,)
        self._check(self.L.rs_reset(self.h, seeds.ctypes.data_as(_up), self._obs.ctypes.data_as(_fp)))
        return self._obs.copy()

    def step(self, actions):
        actions = np.ascontiguousarray(actions, dtype=np.int32).reshape(self.n_envs, self.n_slices)
        self._check(self.L.rs_step(self.h, actions.ctypes.data_as(_ip), self._obs.ctypes.data_as(_fp),
                                   self._reward.ctypes.data_as(_dp), self._labels.ctypes.data_as(_ip),
                                   self._viol.ctypes.data_as(_ip)))
        info = {'SLA_labels': self._labels.copy(), 'violations': self._viol.copy(),
                'total_violations': self._viol.sum(axis=1), 'n_prbs': actions.copy()}
        return self._obs.copy(), self._reward.copy(), np.zeros(self.n_envs, dtype=bool), info

    # ---- device-resident path (bench) --------------------------------------------------
    def random_actions(self, seed, step_index):
        self._check(self.L.rs_random_actions(self.h, int(seed), int(step_index)))

    def step_resident(self):
        self._check(self.L.rs_step_resident(self.h))

    def run_random(self, seed, step_index0, n_steps, graph=False):
        """n_steps x (random_actions(seed, step_index0 + i); step_resident()) enqueued by one call; with
        graph=True the loop body is replayed from a captured hipGraph (same results)."""
        self._check(self.L.rs_run_random(self.h, int(seed), int(step_index0), int(n_steps), 1 if graph else 0))

    def fetch(self):
        actions = np.zeros((self.n_envs, self.n_slices), dtype=np.int32)
        self._check(self.L.rs_fetch(self.h, actions.ctypes.data_as(_ip), self._obs.ctypes.data_as(_fp),
                                    self._reward.ctypes.data_as(_dp), self._labels.ctypes.data_as(_ip),
                                    self._viol.ctypes.data_as(_ip)))
        return dict(actions=actions, obs=self._obs.copy(), reward=self._reward.copy(),
                    labels=self._labels.copy(), violations=self._viol.copy())

    def save_state(self):
        """the handle's whole state as one uint8 array (rs_save_state): feed it to load_state of a handle of the same
        configuration -- this one later, or a fresh one in another process -- and the run goes on bit for bit"""
        n = C.c_uint64()
        self._check(self.L.rs_state_bytes(self.h, C.byref(n)))
        blob = np.empty(n.value, dtype=np.uint8)
        self._check(self.L.rs_save_state(self.h, blob.ctypes.data_as(C.c_void_p), n.value))
        return blob

    def load_state(self, blob):
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        self._check(self.L.rs_load_state(self.h, blob.ctypes.data_as(C.c_void_p), blob.size))

    def synchronize(self):
        self._check(self.L.rs_synchronize(self.h))

    # ---- introspection ------------------------------------------------------------------
    def l1_info(self):
        info = np.zeros((self.n_envs, self.n_ran, 10), dtype=np.float64)   # one row per RAN slice
        self._check(self.L.rs_get_info(self.h, info.ctypes.data_as(_dp)))
        return info

    def set_group_size(self, lanes):
        """lanes per task of the primary step launch (8, 16, 32); results do not depend on it"""
        self._check(self.L.rs_set_group_size(self.h, int(lanes)))

    def set_schedule_hint(self, mode):
        """1: allocations are agent-made (few wide slices), 0: plain instance, -1: automatic; same results"""
        self._check(self.L.rs_set_schedule_hint(self.h, int(mode)))

    def set_alloc_trace(self, enable=True):
        self._check(self.L.rs_set_alloc_trace(self.h, int(bool(enable))))

    def alloc_trace(self):
        shape = ((self.n_envs, self.cfg.slots_per_step, 64) if self.multiplexed   # the one shared UE list; type >> 8 = RAN slice
                 else (self.n_envs, self.cfg.n_embb, self.cfg.slots_per_step, 32))
        tr = np.zeros(shape, dtype=np.dtype(RsAllocRec))
        self._check(self.L.rs_get_alloc_trace(self.h, tr.ctypes.data_as(C.c_void_p)))
        return tr

    def counters(self):
        c = (C.c_uint64 * 4)()
        self._check(self.L.rs_get_counters(self.h, c))
        return [int(x) for x in c]

    def rx_stats(self):
        """(reception tests, of which evaluated the exact probability, short test available) since reset()"""
        c = (C.c_uint64 * 3)()
        self._check(self.L.rs_get_rx_stats(self.h, c))
        return int(c[0]), int(c[1]), bool(c[2])

    def set_kernel_timing(self, enable=True):
        self._check(self.L.rs_set_kernel_timing(self.h, int(bool(enable))))

    def kernel_time_stats_ms(self):
        """((mean, min, max) ms of the dominant step kernel over the launches since the last call, launches)"""
        st = (C.c_double * 3)()
        n = C.c_int64()
        self._check(self.L.rs_kernel_time_stats_ms(self.h, st, C.byref(n)))
        return (st[0], st[1], st[2]), n.value

    def kernel_time_ms(self):
        ms = C.c_double()
        n = C.c_int64()
        self._check(self.L.rs_kernel_time_ms(self.h, C.byref(ms), C.byref(n)))
        return ms.value, n.value
