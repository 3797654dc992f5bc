"""VecKBRL: N independent KBRL agents (one per env replica) on one MI355X.

Batched form of the reference's kbrl_control.KBRL_Control (reference kbrl_control.py:23-157): every
agent keeps its own per-slice Projectron dictionaries on the device; update_control / select_action
run as HIP kernels (network-slicing_amd/csrc/kb_kbrl.hip) behind the kb_* C ABI.
"""
import ctypes as C

import numpy as np

from . import _lib
from .sharding import replica_seeds
from .config import KbConfig, KBRL_ALFA, KBRL_ETA, KBRL_GAMMA

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_fp = C.POINTER(C.c_float)
_up = C.POINTER(C.c_uint64)


class VecKBRL:
    def __init__(self, n_envs, dims, n_prbs, alfa=KBRL_ALFA, accuracy_range=(0.99, 0.999), gamma=KBRL_GAMMA,
                 eta=KBRL_ETA, capacity=4096, device=0, shared=False, first_env=0, pool_bytes=0):
        self.L = _lib.load()
        cfg = KbConfig()
        cfg.n_envs, cfg.n_slices, cfg.n_prbs, cfg.capacity = n_envs, len(dims), n_prbs, capacity
        for i, d in enumerate(dims):
            cfg.dims[i] = int(d)
        cfg.alfa, cfg.acc_lo, cfg.acc_hi = alfa, accuracy_range[0], accuracy_range[1]
        cfg.gamma, cfg.eta = gamma, eta
        cfg.shared_dictionary, cfg.first_env = int(bool(shared)), int(first_env)
        cfg.pool_bytes = int(pool_bytes)
        self.cfg = cfg
        self.n_envs, self.S, self.n_prbs, self.dims = n_envs, len(dims), n_prbs, list(dims)
        self.nv = int(sum(dims))
        self.capacity = capacity
        self.h = C.c_void_p()
        self._check(self.L.kb_create(C.byref(cfg), int(device), C.byref(self.h)))

    def _check(self, rc):
        if rc != 0:
            msg = self.L.kb_last_error(self.h).decode() if self.h else 'kb_create failed'
            raise _lib.RanSliceError(rc, msg)

    def close(self):
        if getattr(self, 'h', None):
            self.L.kb_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self, initial_action, security_factor, seeds=None):
        ia = np.ascontiguousarray(initial_action, dtype=np.int32).reshape(self.n_envs, self.S)
        sf = np.ascontiguousarray(security_factor, dtype=np.int32).reshape(self.n_envs, self.S)
        if seeds is None:
            seeds = replica_seeds(0, self.cfg.first_env, self.n_envs)   # tie-break streams (kernel.py:26-27)
        seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
        self._check(self.L.kb_reset(self.h, ia.ctypes.data_as(_ip), sf.ctypes.data_as(_ip), seeds.ctypes.data_as(_up)))

    def update_control(self, state, action, labels):
        state = np.ascontiguousarray(state, dtype=np.float32).reshape(self.n_envs, self.nv)
        action = np.ascontiguousarray(action, dtype=np.int32).reshape(self.n_envs, self.S)
        labels = np.ascontiguousarray(labels, dtype=np.int32).reshape(self.n_envs, self.S)
        hits = np.zeros((self.n_envs, self.S), dtype=np.int32)
        self._check(self.L.kb_update_control(self.h, state.ctypes.data_as(_fp), action.ctypes.data_as(_ip),
                                             labels.ctypes.data_as(_ip), hits.ctypes.data_as(_ip)))
        return hits

    def select_action(self, state):
        state = np.ascontiguousarray(state, dtype=np.float32).reshape(self.n_envs, self.nv)
        action = np.zeros((self.n_envs, self.S), dtype=np.int32)
        adjusted = np.zeros(self.n_envs, dtype=np.int32)
        self._check(self.L.kb_select_action(self.h, state.ctypes.data_as(_fp), action.ctypes.data_as(_ip),
                                            adjusted.ctypes.data_as(_ip)))
        return action, adjusted

    def _same_library(self, env):
        # an agent handle and an environment handle that meet in one call must come from the same image of the library (the test
        # build's structures need not match the production build's: ranslice._lib.load picks per handle)
        if env.L is not self.L:
            raise RuntimeError('the agent and the environment were created from different builds of libranslice '
                               '(RANSLICE_DEV_BUILD changed between the two constructors?)')

    def step_resident(self, env):
        self._same_library(env)
        self._check(self.L.kb_step_resident(self.h, env.h))

    def run_resident(self, env, n_steps, graph=True):
        """n_steps x (step_resident(env); env.step_resident()) from one call (kb_run_resident); graph: two captured steps
        replayed as a hipGraph -- same results, a fraction of the launches"""
        self._same_library(env)
        self._check(self.L.kb_run_resident(self.h, env.h, int(n_steps), 1 if graph else 0))

    def history_begin(self, steps):
        """start recording KBRL_Control.run's per-step histories on the device (one column per step_resident)"""
        self._hist_steps = int(steps)
        self._check(self.L.kb_history_begin(self.h, int(steps)))

    def history_fetch(self):
        """-> dict of the reference's result arrays per replica: reward f64 [N, steps]; resources, adjusted, SLA,
        violation int16 [N, steps]; hits int16 [N, S, steps] (kbrl_control.py:148-155), and `recorded`"""
        n, st = self.n_envs, self._hist_steps
        sp = C.POINTER(C.c_int16)
        out = dict(reward=np.zeros((n, st)), resources=np.zeros((n, st), dtype=np.int16),
                   hits=np.zeros((n, self.S, st), dtype=np.int16), adjusted=np.zeros((n, st), dtype=np.int16),
                   SLA=np.zeros((n, st), dtype=np.int16), violation=np.zeros((n, st), dtype=np.int16))
        rec = C.c_int32()
        self._check(self.L.kb_history_fetch(self.h, out['reward'].ctypes.data_as(_dp), out['resources'].ctypes.data_as(sp),
                                            out['hits'].ctypes.data_as(sp), out['adjusted'].ctypes.data_as(sp),
                                            out['SLA'].ctypes.data_as(sp), out['violation'].ctypes.data_as(sp),
                                            C.byref(rec)))
        out['recorded'] = rec.value
        return out

    def predict(self, e, s, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y, f = C.c_int32(), C.c_double()
        self._check(self.L.kb_predict(self.h, e, s, x.ctypes.data_as(_dp), C.byref(y), C.byref(f)))
        return y.value, f.value

    def update(self, e, s, x, y):
        x = np.ascontiguousarray(x, dtype=np.float64)
        br, dl = C.c_int32(), C.c_double()
        self._check(self.L.kb_update(self.h, e, s, x.ctypes.data_as(_dp), int(y), C.byref(br), C.byref(dl)))
        return br.value, dl.value

    def learner(self, e, s, with_kinv=False):
        m = C.c_int32()
        self._check(self.L.kb_get_learner(self.h, e, s, C.byref(m), None, None, None))
        m = m.value
        d = self.dims[s] + 1
        L = np.zeros((m, d))
        co = np.zeros(m)
        kinv = np.zeros((m, m)) if with_kinv else None
        if m:
            self._check(self.L.kb_get_learner(self.h, e, s, None, L.ctypes.data_as(_dp), co.ctypes.data_as(_dp),
                                              kinv.ctypes.data_as(_dp) if with_kinv else None))
        return dict(m=m, landmarks=L, coeff=co, kinv=kinv)

    def control(self, with_accuracies=True):
        """margins / security_factors / action [N, S], adjusted [N] and (optionally: it is the big one) the
        accuracy tables [N, S, n_prbs]"""
        T = (self.n_envs, self.S)
        margins, security, action = (np.zeros(T, dtype=np.int32) for _ in range(3))
        adjusted = np.zeros(self.n_envs, dtype=np.int32)
        acc = np.zeros(T + (self.n_prbs,)) if with_accuracies else None
        self._check(self.L.kb_get_control(self.h, margins.ctypes.data_as(_ip), security.ctypes.data_as(_ip),
                                          action.ctypes.data_as(_ip), adjusted.ctypes.data_as(_ip),
                                          acc.ctypes.data_as(_dp) if with_accuracies else None))
        return dict(margins=margins, security_factors=security, action=action, adjusted=adjusted, accuracies=acc)

    def kernel_row(self, e, s):
        """GaussianKernel.k(x) of the last predict(e, s, x) (kernel.py:13-20)"""
        m = C.c_int32()
        self._check(self.L.kb_get_kernel_row(self.h, e, s, C.byref(m), None))
        row = np.zeros(max(m.value, 1))
        if m.value:
            self._check(self.L.kb_get_kernel_row(self.h, e, s, C.byref(m), row.ctypes.data_as(_dp)))
        return row[:m.value] if m.value else np.array([0.0])  # empty dictionary: K_f = [0.0] (projectron.py:35-36)

    def set_adjusted(self, adjusted):
        a = np.ascontiguousarray(adjusted, dtype=np.int32).reshape(self.n_envs)
        self._check(self.L.kb_set_adjusted(self.h, a.ctypes.data_as(_ip)))

    def stats(self):
        s = (C.c_uint64 * 4)()
        self._check(self.L.kb_get_stats(self.h, s))
        return [int(v) for v in s]

    def dictionary_sizes(self):
        """landmarks held by every dictionary: [n_envs, S] (one agent per replica) or [S] (shared)"""
        n = self.S if self.cfg.shared_dictionary else self.n_envs * self.S
        m = np.zeros(n, dtype=np.int32)
        self._check(self.L.kb_get_sizes(self.h, m.ctypes.data_as(_ip)))
        return m if self.cfg.shared_dictionary else m.reshape(self.n_envs, self.S)

    def pool(self):
        """the dictionary pool: dict(used_bytes, total_bytes, saturated = replicas with a dictionary at its capacity,
        pool_full = replicas that found the pool exhausted)"""
        u, t = C.c_uint64(), C.c_uint64()
        ns, nf = C.c_int32(), C.c_int32()
        self._check(self.L.kb_get_pool(self.h, C.byref(u), C.byref(t), C.byref(ns), C.byref(nf)))
        return dict(used_bytes=u.value, total_bytes=t.value, saturated=ns.value, pool_full=nf.value)

    def flagged_replicas(self):
        """replicas whose err word carries 'a dictionary is at its capacity' (bit 8) / 'the pool was exhausted' (bit 16)"""
        e = np.zeros(self.n_envs, dtype=np.int32)
        self._check(self.L.kb_get_flags(self.h, e.ctypes.data_as(_ip)))
        return dict(saturated=[int(k) for k in np.nonzero(e & 8)[0]], pool_full=[int(k) for k in np.nonzero(e & 16)[0]])

    def repair_work(self):
        """bytes the chip-wide repair rounds streamed since reset: dict(matvec_bytes read, rank1_bytes read + written,
        matvec_launches, rank1_launches) -- counted by the kernels from their work plan (kb_get_repair_work)"""
        w = (C.c_uint64 * 8)()
        self._check(self.L.kb_get_repair_work(self.h, w))
        return dict(matvec_bytes=int(w[0]) * (32768 + 1024), rank1_bytes=int(w[1]) * 16384, matvec_launches=int(w[2]),
                    rank1_launches=int(w[3]), direct_passes=int(w[4]), direct_landmarks=int(w[5]))

    def _repair_raw(self):
        w = (C.c_uint64 * 8)()
        self._check(self.L.kb_get_repair_work(self.h, w))
        return list(w)

    def set_kernel_timing(self, enable=True):
        self._check(self.L.kb_set_kernel_timing(self.h, int(bool(enable))))

    def kernel_time_ms(self):
        ms, n = C.c_double(), C.c_int64()
        self._check(self.L.kb_kernel_time_ms(self.h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def phase_times_ms(self):
        """mean device time of the timed launches since the last call: dict(update_ms, select_ms, n_update, n_select) --
        the update phase is update_control_kernel plus the repair kernels"""
        ms = (C.c_double * 2)()
        n = (C.c_int64 * 2)()
        self._check(self.L.kb_phase_times_ms(self.h, ms, n))
        rm = (C.c_double * 2)()
        rn = (C.c_int64 * 2)()
        self._check(self.L.kb_repair_times_ms(self.h, rm, rn))
        km = (C.c_double * 8)()
        kn = (C.c_int64 * 8)()
        self._check(self.L.kb_kernel_times_ms(self.h, km, kn))
        return dict(update_ms=ms[0], select_ms=ms[1], n_update=n[0], n_select=n[1],
                    matvec_launch_ms=rm[0], rank1_launch_ms=rm[1], n_matvec=rn[0], n_rank1=rn[1],
                    select_bin_launch_ms=km[4], n_select_bin=kn[4], finish_launch_ms=km[5], n_finish=kn[5],
                    select_gemm_launch_ms=km[6], n_select_gemm=kn[6], update_small_launch_ms=km[7], n_update_small=kn[7])

    def save_state(self):
        """the handle's whole state as one uint8 array (kb_save_state): feed it to load_state of a handle of the same
        configuration -- this one later, or a fresh one in another process -- and the run goes on bit for bit"""
        n = C.c_uint64()
        self._check(self.L.kb_state_bytes(self.h, C.byref(n)))
        blob = np.empty(n.value, dtype=np.uint8)
        self._check(self.L.kb_save_state(self.h, blob.ctypes.data_as(C.c_void_p), n.value))
        return blob

    def load_state(self, blob):
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        self._check(self.L.kb_load_state(self.h, blob.ctypes.data_as(C.c_void_p), blob.size))
        self._hist_steps = int(np.frombuffer(blob[32:40].tobytes(), dtype=np.uint64)[0])  # kb_state_header.hist_steps

    def synchronize(self):
        self._check(self.L.kb_synchronize(self.h))


PROP_W = 18  # KB_PROP_WIDTH


def merge_proposals(all_counts, all_props, budget):
    """Merge per-rank proposal lists into the list every rank applies: by slice, ascending global replica id,
    truncated to `budget`.  all_counts [W][S], all_props [W][S][budget][PROP_W] -> (counts [S], props
    [S][budget][PROP_W], taken [W][S] = how many of rank w's proposers made it).  Pure host logic (tested on
    CPU with gloo): the result does not depend on how replicas are sharded over ranks."""
    all_counts = np.asarray(all_counts)
    all_props = np.asarray(all_props)
    W, S = all_counts.shape
    out = np.zeros((S, budget, PROP_W))
    counts = np.zeros(S, dtype=np.int32)
    taken = np.zeros((W, S), dtype=np.int32)
    for s in range(S):
        rows = []
        for w in range(W):
            n = min(int(all_counts[w, s]), budget)
            for i in range(n):
                rows.append((all_props[w, s, i, 0], w, i))
        rows.sort()
        rows = rows[:budget]
        for j, (_, w, i) in enumerate(rows):
            out[s, j] = all_props[w, s, i]
            taken[w, s] += 1
        counts[s] = len(rows)
    return counts, out, taken


class SharedVecKBRL(VecKBRL):
    """One KBRL dictionary per slice index, learned from every replica of every rank (build-defined
    extension; the reference trains one independent agent per run).

    Default: the whole learning step runs on the device (kb_shared_step) -- scan, all-gather of the proposal blocks
    over RCCL (bound directly by libranslice.so, on the agent's HIP stream), merge, apply, commit.  One process per
    GPU joins the communicator with `comm_init(unique_id, rank, world)`; `SharedVecKBRL.unique_id()` is generated by
    one rank and distributed by whatever launched the ranks.  Without comm_init the handle is its own world.

    `exchange(counts, props)` (optional) replaces the device exchange by a host callback that must return the lists of
    all ranks stacked on a leading axis ([W][S], [W][S][budget][PROP_W]) and this rank's index: used to run several
    handles side by side in ONE process (tests) and to pin the merge rule on CPU (merge_proposals, gloo test)."""

    def __init__(self, n_envs, dims, n_prbs, budget=64, max_rounds=4, exchange=None, **kw):
        super().__init__(n_envs, dims, n_prbs, shared=True, **kw)
        self.budget, self.max_rounds = budget, max_rounds
        self.exchange = exchange
        self.rounds_last = 0

    @staticmethod
    def unique_id():
        """128 opaque bytes (ncclGetUniqueId) for comm_init; call on ONE rank"""
        buf = C.create_string_buffer(128)
        rc = _lib.load().kb_comm_unique_id(buf)
        if rc != 0:
            raise _lib.RanSliceError(rc, 'kb_comm_unique_id failed (librccl.so missing?)')
        return buf.raw

    def comm_info(self):
        """(rank, world) as the live RCCL communicator reports them; (0, 1) without one"""
        r, w = C.c_int(), C.c_int()
        self._check(self.L.kb_comm_info(self.h, C.byref(r), C.byref(w)))
        return r.value, w.value

    def comm_init(self, unique_id, rank, world):
        if self.cfg.first_env != rank * self.n_envs:
            raise ValueError('first_env must be rank * n_envs (contiguous replica shards)')
        buf = C.create_string_buffer(bytes(unique_id), 128)
        self._check(self.L.kb_comm_init(self.h, buf, int(rank), int(world)))

    def step_resident(self, env, count_rounds=False):
        """one closed-loop agent step on the device: the shared learning step on the simulator's buffers, then
        select_action into them (kb_shared_step_resident).  count_rounds: wait for the number of exchange rounds
        that had proposals (rounds_last); without it the host does not wait for the last permitted round at all"""
        self._same_library(env)
        if count_rounds:
            rounds = C.c_int32()
            self._check(self.L.kb_shared_step_resident(self.h, env.h, self.budget, self.max_rounds, C.byref(rounds)))
            self.rounds_last = rounds.value
        else:
            self._check(self.L.kb_shared_step_resident(self.h, env.h, self.budget, self.max_rounds, None))
            self.rounds_last = None

    def update_control(self, state, action, labels):
        state = np.ascontiguousarray(state, dtype=np.float32).reshape(self.n_envs, self.nv)
        action = np.ascontiguousarray(action, dtype=np.int32).reshape(self.n_envs, self.S)
        labels = np.ascontiguousarray(labels, dtype=np.int32).reshape(self.n_envs, self.S)
        hits = np.zeros((self.n_envs, self.S), dtype=np.int32)
        if self.exchange is None:
            rounds = C.c_int32()
            self._check(self.L.kb_shared_step(self.h, state.ctypes.data_as(_fp), action.ctypes.data_as(_ip),
                                              labels.ctypes.data_as(_ip), self.budget, self.max_rounds,
                                              hits.ctypes.data_as(_ip), C.byref(rounds)))
            self.rounds_last = rounds.value
            return hits
        counts = np.zeros(self.S, dtype=np.int32)
        props = np.zeros((self.S, self.budget, PROP_W))
        self.rounds_last = 0
        for rnd in range(self.max_rounds):
            first = rnd == 0
            self._check(self.L.kb_shared_scan(
                self.h, state.ctypes.data_as(_fp) if first else None, action.ctypes.data_as(_ip) if first else None,
                labels.ctypes.data_as(_ip) if first else None, rnd, self.budget, hits.ctypes.data_as(_ip),
                counts.ctypes.data_as(_ip), props.ctypes.data_as(_dp)))
            all_counts, all_props, me = self.exchange(counts, props)
            if int(np.asarray(all_counts).sum()) == 0:
                break
            mc, mp, taken = merge_proposals(all_counts, all_props, self.budget)
            mc = np.ascontiguousarray(mc, dtype=np.int32)
            mp = np.ascontiguousarray(mp, dtype=np.float64)
            self._check(self.L.kb_shared_apply(self.h, mc.ctypes.data_as(_ip), mp.ctypes.data_as(_dp), self.budget))
            acc = np.ascontiguousarray(taken[me], dtype=np.int32)
            self._check(self.L.kb_shared_commit(self.h, acc.ctypes.data_as(_ip)))
            self.rounds_last = rnd + 1
        return hits
