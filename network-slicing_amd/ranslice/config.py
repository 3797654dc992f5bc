"""Scenario constants and the ctypes mirror of `rs_config` (include/ranslice.h).

All numbers are the reference's module-level constants; nothing is tuned here.
  scenarios / traffic / SLA / state variables : reference scenario_creator.py:26-96
  normalisation constants                      : reference scenario_creator.py:115-134
  radio constants, MI sigmoid parameters       : reference channel_models.py:21-27, 121-124, 268-270
  PF scheduler parameters                      : reference schedulers.py:13
  MCS table                                    : reference datasets/mcs_codeset.csv (columns rate, snr,
                                                 order, modulation are the ones read, channel_models.py:262-265)
"""
import ctypes as C

RS_MAX_MCS = 32
RS_MAX_SET = 8
N_EMBB_VARS = 10
N_MMTC_VARS = 3

# scenario_creator.py:26-50 (list index -> dict; the dict names there are off by one)
SCENARIOS = (
    dict(n_prbs=200, n_embb=5, n_mmtc=0),
    dict(n_prbs=150, n_embb=3, n_mmtc=2),
    dict(n_prbs=100, n_embb=1, n_mmtc=4),
    dict(n_prbs=70, n_embb=1, n_mmtc=1),
)

CBR_DESCRIPTION = {'lambda': 2.0 / 60.0, 't_mean': 30.0, 'bit_rate': 500000}
VBR_DESCRIPTION = {'lambda': 5.0 / 60.0, 't_mean': 30.0, 'p_size': 1000, 'b_size': 500, 'b_rate': 1}
SLA_EMBB = {'cbr_th': 10e6, 'cbr_prb': 20, 'cbr_queue': 10e4, 'vbr_th': 15e6, 'vbr_prb': 30, 'vbr_queue': 15e4}
STATE_VARIABLES_EMBB = ['cbr_traffic', 'cbr_th', 'cbr_prb', 'cbr_queue', 'cbr_snr',
                        'vbr_traffic', 'vbr_th', 'vbr_prb', 'vbr_queue', 'vbr_snr']
MTC_DESCRIPTION = {
    'n_devices': 1000,
    'repetition_set': [2, 4, 8, 16, 32, 64, 128],
    'period_set': [1000, 50000, 10000, 15000, 20000, 25000, 50000, 100000],
}
STATE_VARIABLES_MMTC = ['devices', 'avg_rep', 'delay']
SLA_MMTC = {'delay': 300}

PROPAGATION = {  # channel_models.py:121-124
    'macro_cell_urban_2GHz': (128.1, 37.6),
    'macro_cell_urban_900MHz': (120.9, 37.6),
    'macro_cell_rural': (95.5, 34.1),
}

MI_PARAMETERS = {'qpsk': (-0.25040431, 0.31591749), '16qam': (5.12440916, 0.25423209),
                 '64qam': (9.16962738, 0.22298101)}
_MOD_ID = {'qpsk': 0, '16qam': 1, '64qam': 2}

# datasets/mcs_codeset.csv: (rate, snr, order, modulation)
MCS_TABLE = (
    (0.2, -2.7, 2, 'qpsk'), (0.25, -1.3, 2, 'qpsk'), (0.333333333, -0.8, 2, 'qpsk'), (0.4, -0.2, 2, 'qpsk'),
    (0.5, 1.3, 2, 'qpsk'), (0.6, 2.7, 2, 'qpsk'), (0.666666667, 3.4, 2, 'qpsk'), (0.75, 4.6, 2, 'qpsk'),
    (0.4, 5.3, 4, '16qam'), (0.45, 6.2, 4, '16qam'), (0.5, 6.8, 4, '16qam'), (0.55, 7.8, 4, '16qam'),
    (0.6, 8.7, 4, '16qam'), (0.666666667, 9.3, 4, '16qam'), (0.75, 10.7, 4, '16qam'), (0.8, 11.2, 4, '16qam'),
    (0.833333333, 12.2, 4, '16qam'), (0.6, 13.6, 6, '64qam'), (0.625, 14.0, 6, '64qam'),
    (0.666666667, 14.5, 6, '64qam'), (0.708333333, 15.4, 6, '64qam'), (0.75, 16.3, 6, '64qam'),
    (0.8, 16.8, 6, '64qam'), (0.833333333, 17.8, 6, '64qam'), (0.875, 18.6, 6, '64qam'), (0.9, 19.2, 6, '64qam'),
)


class RsConfig(C.Structure):
    _fields_ = [
        ('n_envs', C.c_int32), ('n_prbs', C.c_int32), ('n_embb', C.c_int32), ('n_mmtc', C.c_int32),
        ('slots_per_step', C.c_int32), ('max_ue', C.c_int32), ('max_bursts', C.c_int32),
        ('max_mtc_queue', C.c_int32),
        ('slot_length', C.c_double), ('penalty', C.c_double),
        ('cbr_lambda', C.c_double), ('cbr_t_mean', C.c_double), ('cbr_bit_rate', C.c_double),
        ('vbr_lambda', C.c_double), ('vbr_t_mean', C.c_double), ('vbr_p_size', C.c_double),
        ('vbr_b_size', C.c_double), ('vbr_b_rate', C.c_double),
        ('sla_embb', C.c_double * 6),
        ('norm_embb', C.c_double * N_EMBB_VARS),
        ('mtc_n_devices', C.c_int32), ('mtc_n_rep', C.c_int32), ('mtc_n_period', C.c_int32),
        ('mtc_rep_set', C.c_int32 * RS_MAX_SET), ('mtc_period_set', C.c_int32 * RS_MAX_SET),
        ('sla_mtc_delay', C.c_double),
        ('norm_mmtc', C.c_double * N_MMTC_VARS),
        ('prop_A', C.c_double), ('prop_B', C.c_double),
        ('pf_granularity', C.c_int32), ('pf_window', C.c_int32), ('sym_per_prb', C.c_int32),
        ('n_mcs', C.c_int32),
        ('mcs_rate', C.c_double * RS_MAX_MCS), ('mcs_snr', C.c_double * RS_MAX_MCS),
        ('mcs_order', C.c_int32 * RS_MAX_MCS), ('mcs_mod', C.c_int32 * RS_MAX_MCS),
        ('mi_x0', C.c_double * 3), ('mi_k', C.c_double * 3),
        ('l1_multiplex', C.c_int32), ('reserved_', C.c_int32),
    ]


KB_MAX_SLICES = 8

# KBRL learner initialisation (reference scenario_creator.py:185-193, 218; projectron.py:25)
KBRL_ALFA = 0.05
EMBB_SEC, EMBB_A = (2, 8), (4, 20)
MMTC_SEC, MMTC_A = (1, 4), (2, 10)
KBRL_GAMMA, KBRL_ETA = 1.0, 0.1


class KbConfig(C.Structure):
    _fields_ = [('n_envs', C.c_int32), ('n_slices', C.c_int32), ('n_prbs', C.c_int32), ('capacity', C.c_int32),
                ('dims', C.c_int32 * KB_MAX_SLICES), ('alfa', C.c_double), ('acc_lo', C.c_double),
                ('acc_hi', C.c_double), ('gamma', C.c_double), ('eta', C.c_double),
                ('shared_dictionary', C.c_int32), ('first_env', C.c_int32), ('pool_bytes', C.c_int64)]


class RsAllocRec(C.Structure):
    _fields_ = [('serial', C.c_int32), ('type', C.c_int32), ('e_snr', C.c_int32), ('prbs', C.c_int32),
                ('bits', C.c_int64), ('queue', C.c_double), ('th', C.c_double), ('p', C.c_double)]


def make_config(scenario, n_envs=1, slots_per_step=50, propagation_type='macro_cell_urban_2GHz', penalty=100,
                n_prbs=None, n_embb=None, n_mmtc=None, max_ue=0, max_bursts=0, max_mtc_queue=0, L1_level=True):
    """Build the rs_config for scenario index `scenario` exactly as create_env would wire it
    (reference scenario_creator.py:100-183)."""
    sc = dict(SCENARIOS[scenario]) if scenario is not None else {}
    if n_prbs is not None:
        sc['n_prbs'] = n_prbs
    if n_embb is not None:
        sc['n_embb'] = n_embb
    if n_mmtc is not None:
        sc['n_mmtc'] = n_mmtc
    cfg = RsConfig()
    cfg.n_envs = n_envs
    cfg.n_prbs, cfg.n_embb, cfg.n_mmtc = sc['n_prbs'], sc['n_embb'], sc['n_mmtc']
    cfg.slots_per_step = slots_per_step
    cfg.max_ue, cfg.max_bursts, cfg.max_mtc_queue = max_ue, max_bursts, max_mtc_queue
    cfg.l1_multiplex = 0 if L1_level else 1   # create_env(..., L1_level) (scenario_creator.py:156-177)
    cfg.slot_length = 1e-3
    cfg.penalty = penalty
    cfg.cbr_lambda, cfg.cbr_t_mean, cfg.cbr_bit_rate = (CBR_DESCRIPTION['lambda'], CBR_DESCRIPTION['t_mean'],
                                                        CBR_DESCRIPTION['bit_rate'])
    cfg.vbr_lambda, cfg.vbr_t_mean = VBR_DESCRIPTION['lambda'], VBR_DESCRIPTION['t_mean']
    cfg.vbr_p_size, cfg.vbr_b_size, cfg.vbr_b_rate = (VBR_DESCRIPTION['p_size'], VBR_DESCRIPTION['b_size'],
                                                     VBR_DESCRIPTION['b_rate'])
    for i, k in enumerate(('cbr_th', 'cbr_prb', 'cbr_queue', 'vbr_th', 'vbr_prb', 'vbr_queue')):
        cfg.sla_embb[i] = SLA_EMBB[k]
    time_per_step = slots_per_step * 1e-3  # scenario_creator.py:106
    norm_embb = {  # scenario_creator.py:115-126
        'cbr_traffic': 5e6 * time_per_step, 'cbr_th': 10e6 * time_per_step, 'cbr_prb': 25 * slots_per_step,
        'cbr_queue': 10e4 * slots_per_step, 'cbr_snr': 35 * slots_per_step,
        'vbr_traffic': 5e6 * time_per_step, 'vbr_th': 10e6 * time_per_step, 'vbr_prb': 35 * slots_per_step,
        'vbr_queue': 10e4 * slots_per_step, 'vbr_snr': 35 * slots_per_step,
    }
    for i, k in enumerate(STATE_VARIABLES_EMBB):
        cfg.norm_embb[i] = norm_embb[k]
    cfg.mtc_n_devices = MTC_DESCRIPTION['n_devices']
    cfg.mtc_n_rep = len(MTC_DESCRIPTION['repetition_set'])
    cfg.mtc_n_period = len(MTC_DESCRIPTION['period_set'])
    for i, v in enumerate(MTC_DESCRIPTION['repetition_set']):
        cfg.mtc_rep_set[i] = v
    for i, v in enumerate(MTC_DESCRIPTION['period_set']):
        cfg.mtc_period_set[i] = v
    cfg.sla_mtc_delay = SLA_MMTC['delay']
    for i in range(3):  # scenario_creator.py:130-134: devices, avg_rep, delay all 100*slots
        cfg.norm_mmtc[i] = 100 * slots_per_step
    cfg.prop_A, cfg.prop_B = PROPAGATION[propagation_type]
    cfg.pf_granularity, cfg.pf_window, cfg.sym_per_prb = 2, 50, 158
    cfg.n_mcs = len(MCS_TABLE)
    for i, (rate, snr, order, mod) in enumerate(MCS_TABLE):
        cfg.mcs_rate[i], cfg.mcs_snr[i], cfg.mcs_order[i], cfg.mcs_mod[i] = rate, snr, order, _MOD_ID[mod]
    for name, m in _MOD_ID.items():
        cfg.mi_x0[m], cfg.mi_k[m] = MI_PARAMETERS[name]
    return cfg


def n_vars(cfg):
    return cfg.n_embb * N_EMBB_VARS + cfg.n_mmtc * N_MMTC_VARS
