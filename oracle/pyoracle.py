"""ctypes binding of the CPU oracle (oracle/rs_oracle.c, oracle/kb_oracle.c).

TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this module; the product (network-slicing_amd/) never does.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
_PKG = os.path.join(_ROOT, 'network-slicing_amd')
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)

from ranslice.config import RsConfig, RsAllocRec, n_vars  # noqa: E402

LIB_PATH = os.path.join(_HERE, 'build', 'librs_oracle.so')


def build(force=False):
    if force or not os.path.exists(LIB_PATH):
        subprocess.check_call(['make', '-s', '-C', _HERE] + (['-B'] if force else []))
    return LIB_PATH


_lib = None

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_lp = C.POINTER(C.c_int64)


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        L.rso_create.restype = C.c_void_p
        L.rso_create.argtypes = [C.POINTER(RsConfig)]
        L.rso_destroy.argtypes = [C.c_void_p]
        L.rso_load_fading.argtypes = [C.c_void_p, C.c_int, _dp, C.c_int, C.c_int]
        L.rso_set_tape.argtypes = [C.c_void_p, C.POINTER(C.c_uint8), _dp, C.c_int64]
        L.rso_tape_pos.restype = C.c_int64
        L.rso_tape_pos.argtypes = [C.c_void_p]
        L.rso_set_seed.argtypes = [C.c_void_p, C.c_uint64]
        L.rso_reset.argtypes = [C.c_void_p]
        L.rso_step.argtypes = [C.c_void_p, _ip, C.POINTER(C.c_float), _dp, _ip, _ip, _dp, C.c_void_p]
        L.rso_error.restype = C.c_char_p
        L.rso_error.argtypes = [C.c_void_p]
        L.rso_get_counters.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.rso_max_ue.argtypes = [C.c_void_p]
        L.rso_get_mtc_queue.argtypes = [C.c_void_p, C.c_int, _lp, _lp, C.c_int, _lp]
        L.rso_random_actions.argtypes = [C.POINTER(RsConfig), C.c_uint64, C.c_uint64, C.c_int64, _ip]
        L.rso_bench_run.restype = C.c_double
        L.rso_bench_run.argtypes = [C.c_void_p, C.c_uint64, C.c_int64, C.c_uint64, C.c_int64]
        L.rso_mcs_factors.argtypes = [_dp, _dp]
        L.rso_mcs_lookup.argtypes = [C.POINTER(RsConfig), C.c_int, _ip, _ip]
        L.rso_response.restype = C.c_double
        L.rso_response.argtypes = [C.POINTER(RsConfig), C.c_int, _dp, C.c_int]
        L.rso_pairwise_sum.restype = C.c_double
        L.rso_pairwise_sum.argtypes = [_dp, C.c_int64]
        L.rso_pf_allocate.argtypes = [C.POINTER(RsConfig), C.c_int, C.c_int, _dp, _dp, _ip, _dp, _lp, _lp, _dp]
        L.rso_vbr_source.argtypes = [C.POINTER(RsConfig), _dp, C.c_int64, C.c_int64, _dp, _lp]
        L.rso_macro_cell.restype = C.c_double
        L.rso_macro_cell.argtypes = [C.POINTER(RsConfig), _dp, C.c_int, C.c_double, _ip]
        for f in ('rso_exp', 'rso_log', 'rso_acos'):
            getattr(L, f).restype = C.c_double
            getattr(L, f).argtypes = [C.c_double]
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(t)


class OracleError(RuntimeError):
    pass


class OracleEnv:
    """One env replica of the oracle."""

    def __init__(self, cfg, fading=None):
        self.cfg = cfg
        self.L = lib()
        self.h = self.L.rso_create(C.byref(cfg))
        self.n_ran = cfg.n_embb + cfg.n_mmtc
        # action / label entries: one per L1 slice (all eMBB RAN slices share one, all mMTC ones another, when multiplexed)
        self.n_slices = ((cfg.n_embb > 0) + (cfg.n_mmtc > 0)) if cfg.l1_multiplex else self.n_ran
        self.n_vars = n_vars(cfg)
        self.max_ue = self.L.rso_max_ue(self.h)
        self._keep = []
        if fading is not None:
            for t, tab in enumerate(fading):
                self.load_fading(t, tab)

    def __del__(self):
        if getattr(self, 'h', None):
            self.L.rso_destroy(self.h)
            self.h = None

    def _check(self, rc):
        if rc != 0:
            raise OracleError('oracle error %d: %s' % (rc, self.L.rso_error(self.h).decode()))

    def load_fading(self, trace_id, table):
        table = np.ascontiguousarray(table, dtype=np.float64)
        self._check(self.L.rso_load_fading(self.h, trace_id, _p(table, _dp), table.shape[0], table.shape[1]))

    def set_tape(self, kind, val):
        kind = np.ascontiguousarray(kind, dtype=np.uint8)
        val = np.ascontiguousarray(val, dtype=np.float64)
        self._keep = [kind, val]
        self.L.rso_set_tape(self.h, _p(kind, C.POINTER(C.c_uint8)), _p(val, _dp), len(kind))

    def tape_pos(self):
        return self.L.rso_tape_pos(self.h)

    def set_seed(self, seed):
        self.L.rso_set_seed(self.h, int(seed))

    def reset(self):
        self._check(self.L.rso_reset(self.h))
        return np.zeros(self.n_vars, dtype=np.float32)

    def step(self, action, trace=False):
        action = np.ascontiguousarray(action, dtype=np.int32)
        obs = np.zeros(self.n_vars, dtype=np.float32)
        reward = np.zeros(1, dtype=np.float64)
        labels = np.zeros(self.n_slices, dtype=np.int32)
        viol = np.zeros(self.n_slices, dtype=np.int32)
        info = np.zeros((self.n_ran, 10), dtype=np.float64)
        tr = None
        trp = None
        if trace:
            n_l1_embb = min(self.cfg.n_embb, 1) if self.cfg.l1_multiplex else self.cfg.n_embb
            tr = np.zeros((n_l1_embb, self.cfg.slots_per_step, self.max_ue), dtype=np.dtype(RsAllocRec))
            trp = tr.ctypes.data_as(C.c_void_p)
        self._check(self.L.rso_step(self.h, _p(action, _ip), _p(obs, C.POINTER(C.c_float)), _p(reward, _dp),
                                    _p(labels, _ip), _p(viol, _ip), _p(info, _dp), trp))
        out = dict(obs=obs, reward=float(reward[0]), labels=labels, violations=viol, info=info)
        if trace:
            out['trace'] = tr
        return out

    def mtc_queue(self, s, cap=4096):
        """(time, remaining repetitions, arrival times) of mMTC slice s's FIFO, head first"""
        rep = np.zeros(cap, dtype=np.int64)
        start = np.zeros(cap, dtype=np.int64)
        t = np.zeros(1, dtype=np.int64)
        n = self.L.rso_get_mtc_queue(self.h, int(s), _p(rep, _lp), _p(start, _lp), cap, _p(t, _lp))
        assert 0 <= n <= cap
        return int(t[0]), rep[:n].copy(), start[:n].copy()

    def bench_run(self, action_seed, replica, step0, n_steps):
        return self.L.rso_bench_run(self.h, int(action_seed), int(replica), int(step0), int(n_steps))

    def counters(self):
        c = (C.c_uint64 * 4)()
        self.L.rso_get_counters(self.h, c)
        return [int(x) for x in c]


def random_actions(cfg, seed, step_index, replica):
    a = np.zeros(cfg.n_embb + cfg.n_mmtc, dtype=np.int32)
    lib().rso_random_actions(C.byref(cfg), int(seed), int(step_index), int(replica), _p(a, _ip))
    return a


def mcs_factors():
    a, b = C.c_double(), C.c_double()
    lib().rso_mcs_factors(C.byref(a), C.byref(b))
    return a.value, b.value


def mcs_lookup(cfg, e_snr):
    m, r = C.c_int32(), C.c_int32()
    lib().rso_mcs_lookup(C.byref(cfg), int(e_snr), C.byref(m), C.byref(r))
    return m.value, r.value


def response(cfg, mcs, snr):
    snr = np.ascontiguousarray(snr, dtype=np.float64)
    return lib().rso_response(C.byref(cfg), int(mcs), _p(snr, _dp), len(snr))


def pairwise_sum(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return lib().rso_pairwise_sum(_p(a, _dp), len(a))


def pf_allocate(cfg, th, queue, e_snr, snr):
    th = np.ascontiguousarray(th, dtype=np.float64)
    queue = np.ascontiguousarray(queue, dtype=np.float64)
    e_snr = np.ascontiguousarray(e_snr, dtype=np.int32)
    snr = np.ascontiguousarray(snr, dtype=np.float64)
    n_ue, n_prb = snr.shape
    prbs = np.zeros(n_ue, dtype=np.int64)
    bits = np.zeros(n_ue, dtype=np.int64)
    p = np.zeros(n_ue, dtype=np.float64)
    lib().rso_pf_allocate(C.byref(cfg), n_ue, n_prb, _p(th, _dp), _p(queue, _dp), _p(e_snr, _ip), _p(snr, _dp),
                          _p(prbs, _lp), _p(bits, _lp), _p(p, _dp))
    return prbs, bits, p


def vbr_source(cfg, gexp, n_slots):
    gexp = np.ascontiguousarray(gexp, dtype=np.float64)
    bits = np.zeros(n_slots, dtype=np.float64)
    used = C.c_int64()
    lib().rso_vbr_source(C.byref(cfg), _p(gexp, _dp), len(gexp), n_slots, _p(bits, _dp), C.byref(used))
    return bits, used.value


def macro_cell(cfg, uv, normal):
    uv = np.ascontiguousarray(uv, dtype=np.float64)
    used = C.c_int32()
    v = lib().rso_macro_cell(C.byref(cfg), _p(uv, _dp), len(uv), float(normal), C.byref(used))
    return v, used.value


# ----------------------------------------------------------------------------- KBRL oracle
def _kb_lib():
    L = lib()
    if not getattr(L, '_kb_ready', False):
        L.kbo_create.restype = C.c_void_p
        L.kbo_create.argtypes = [C.c_int, _ip, C.c_int, C.c_double, C.c_double, C.c_double, _ip, _ip, C.c_double,
                                 C.c_double, C.c_int]
        L.kbo_destroy.argtypes = [C.c_void_p]
        L.kbo_set_tape.argtypes = [C.c_void_p, _dp, C.c_int64]
        L.kbo_set_seed.argtypes = [C.c_void_p, C.c_uint64]
        L.kbo_tape_pos.restype = C.c_int64
        L.kbo_tape_pos.argtypes = [C.c_void_p]
        L.kbo_predict.argtypes = [C.c_void_p, C.c_int, _dp, _dp]
        L.kbo_update.argtypes = [C.c_void_p, C.c_int, _dp, C.c_int, _dp]
        L.kbo_set_size.argtypes = [C.c_void_p, C.c_int]
        L.kbo_m.argtypes = [C.c_void_p, C.c_int]
        for f in ('kbo_coeff', 'kbo_landmarks'):
            getattr(L, f).restype = _dp
            getattr(L, f).argtypes = [C.c_void_p, C.c_int]
        L.kbo_kinv.restype = _dp
        L.kbo_kinv.argtypes = [C.c_void_p, C.c_int, _ip]
        L.kbo_select_action.argtypes = [C.c_void_p, C.POINTER(C.c_float), _ip]
        L.kbo_update_control.argtypes = [C.c_void_p, C.POINTER(C.c_float), _ip, _ip, C.c_int, _ip]
        L.kbo_margins.restype = _ip
        L.kbo_margins.argtypes = [C.c_void_p]
        L.kbo_security_factors.restype = _ip
        L.kbo_security_factors.argtypes = [C.c_void_p]
        L.kbo_accuracies.restype = _dp
        L.kbo_accuracies.argtypes = [C.c_void_p]
        L.kbo_n_predict.restype = C.c_int64
        L.kbo_n_predict.argtypes = [C.c_void_p]
        L.kbo_n_mistakes.restype = C.c_int64
        L.kbo_n_mistakes.argtypes = [C.c_void_p]
        L.kbo_error.argtypes = [C.c_void_p]
        L._kb_ready = True
    return L


class OracleKBRL:
    """The reference's KBRL_Control (one agent) restated in C."""

    def __init__(self, dims, n_prbs, initial_action, security_factor, alfa=0.05, accuracy_range=(0.99, 0.999),
                 gamma=1.0, eta=0.1, capacity=1024):
        self.L = _kb_lib()
        self.dims = np.ascontiguousarray(dims, dtype=np.int32)
        self.S = len(self.dims)
        self.n_prbs = n_prbs
        ia = np.ascontiguousarray(initial_action, dtype=np.int32)
        sf = np.ascontiguousarray(security_factor, dtype=np.int32)
        self.cap = capacity
        self.h = self.L.kbo_create(self.S, _p(self.dims, _ip), n_prbs, alfa, accuracy_range[0], accuracy_range[1],
                                   _p(ia, _ip), _p(sf, _ip), gamma, eta, capacity)
        self.action = ia.copy()
        self.adjusted = 0
        self._keep = None

    def __del__(self):
        if getattr(self, 'h', None):
            self.L.kbo_destroy(self.h)
            self.h = None

    def set_tape(self, val):
        val = np.ascontiguousarray(val, dtype=np.float64)
        self._keep = val
        self.L.kbo_set_tape(self.h, _p(val, _dp), len(val))

    def set_seed(self, seed):
        self.L.kbo_set_seed(self.h, int(seed))

    def predict(self, s, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        f = C.c_double()
        y = self.L.kbo_predict(self.h, s, _p(x, _dp), C.byref(f))
        return y, f.value

    def update(self, s, x, y):
        x = np.ascontiguousarray(x, dtype=np.float64)
        d = C.c_double(float('nan'))
        br = self.L.kbo_update(self.h, s, _p(x, _dp), int(y), C.byref(d))
        return br, d.value

    def m(self, s):
        return self.L.kbo_m(self.h, s)

    def set_size(self, s):
        return self.L.kbo_set_size(self.h, s)

    def coeff(self, s):
        return np.ctypeslib.as_array(self.L.kbo_coeff(self.h, s), shape=(self.cap,))[:self.m(s)].copy()

    def landmarks(self, s):
        d = int(self.dims[s]) + 1
        return np.ctypeslib.as_array(self.L.kbo_landmarks(self.h, s), shape=(self.cap, d))[:self.m(s)].copy()

    def kinv(self, s):
        ld = C.c_int32()
        p = self.L.kbo_kinv(self.h, s, C.byref(ld))
        m = self.m(s)
        return np.ctypeslib.as_array(p, shape=(ld.value, ld.value))[:m, :m].copy()

    def select_action(self, state):
        state = np.ascontiguousarray(state, dtype=np.float32)
        act = np.zeros(self.S, dtype=np.int32)
        adj = self.L.kbo_select_action(self.h, _p(state, C.POINTER(C.c_float)), _p(act, _ip))
        self.action = act
        return act, adj

    def update_control(self, state, action, labels):
        state = np.ascontiguousarray(state, dtype=np.float32)
        action = np.ascontiguousarray(action, dtype=np.int32)
        labels = np.ascontiguousarray(labels, dtype=np.int32)
        hits = np.zeros(self.S, dtype=np.int32)
        self.L.kbo_update_control(self.h, _p(state, C.POINTER(C.c_float)), _p(action, _ip), _p(labels, _ip),
                                  int(self.adjusted), _p(hits, _ip))
        return hits

    @property
    def margins(self):
        return np.ctypeslib.as_array(self.L.kbo_margins(self.h), shape=(self.S,)).copy()

    @property
    def security_factors(self):
        return np.ctypeslib.as_array(self.L.kbo_security_factors(self.h), shape=(self.S,)).copy()

    @property
    def accuracies(self):
        return np.ctypeslib.as_array(self.L.kbo_accuracies(self.h), shape=(self.S, self.n_prbs)).copy()

    def stats(self):
        return self.L.kbo_n_predict(self.h), self.L.kbo_n_mistakes(self.h)

    def error(self):
        return self.L.kbo_error(self.h)
