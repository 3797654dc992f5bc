"""G9-G11: golden sequences for the KBRL agent, recorded from the reference (see gen_golden.py)."""
import numpy as np


def _g9(rh, tape, save):
    from algorithms.kernel import GaussianKernel
    from algorithms.projectron import SVvariable, Projectron
    out = {}
    for tag, d, n in (('d11', 11, 1500), ('d4', 4, 1200)):
        rng = np.random.default_rng(90 + d)
        np.random.seed(9)
        tape.clear()
        alg = Projectron(GaussianKernel(SVvariable(), 1))
        xs, ys, fs, ypred, branch, delta, ms = [], [], [], [], [], [], []
        for i in range(n):
            x = np.append(rng.random(d - 1).astype(np.float32), rng.integers(0, 201) / 200)
            score = x[:-1].mean() * 0.8 + 0.35 - x[-1]
            y = -1 if score + rng.normal(0, 0.05) > 0 else 1
            if i in (3, 4):  # exercise f == 0 ties: a far-away point whose kernel values underflow in f32
                x = x + (30.0 if i == 3 else 60.0)
            yp = alg.predict(x)
            f = float(alg.f)
            m0 = alg.counter
            dl = np.nan
            if alg.f * y <= 0:  # the expression of projectron.py:41-44, evaluated on the reference's arrays
                d_star = alg.Kinv @ alg.K_f
                if np.ndim(d_star) == 0:
                    d_star = np.array([d_star], dtype=np.float32)
                dl = float(max(alg.kernel.k_eval(x, x) - d_star @ alg.K_f, 0))
            alg.update(x, y)
            br = 0 if not (f * y <= 0) else (2 if alg.counter > m0 else 1)
            xs.append(x); ys.append(y); fs.append(f); ypred.append(int(yp)); branch.append(br)
            delta.append(dl); ms.append(alg.counter)
        kind, val = tape.arrays()
        out.update({tag + '_x': np.asarray(xs), tag + '_y': np.asarray(ys, dtype=np.int32),
                    tag + '_f': np.asarray(fs), tag + '_ypred': np.asarray(ypred, dtype=np.int32),
                    tag + '_branch': np.asarray(branch, dtype=np.int32), tag + '_delta': np.asarray(delta),
                    tag + '_m': np.asarray(ms, dtype=np.int32), tag + '_ties': val,
                    tag + '_landmarks': np.atleast_2d(alg.sv.landmarks), tag + '_coeff': np.asarray(alg.sv.coeff, dtype=np.float64),
                    tag + '_kinv': np.atleast_2d(np.asarray(alg.Kinv, dtype=np.float64)),
                    tag + '_set_size': np.int32(alg.get_set_size())})
    save('g9_projectron', **out)


def _run_agent(rh, tape, scenario, seed, steps, a_range):
    import scenario_creator as sc
    np.random.seed(2000 + seed)
    rng = rh.TapeRNG(np.random.default_rng(seed), tape)
    env = sc.create_env(rng, scenario)
    tape.on = False
    agent = sc.create_kbrl_agent(rng, scenario, accuracy_range=a_range)
    tape.on = True
    init_action = agent.action.copy()
    init_sec = agent.security_factors.copy()
    rec = dict(state=[], action_in=[], labels=[], hits=[], action_out=[], adjusted=[], margins=[], security=[],
               acc=[], set_size=[], reward=[], violation=[])
    tape.clear()
    action = agent.action
    state = env.reset()
    for i in range(steps):
        new_state, reward, _, info = env.step(action)
        labels = info['SLA_labels']
        rec['state'].append(np.array(state, dtype=np.float32)); rec['action_in'].append(np.array(action, dtype=np.int32))
        rec['labels'].append(np.array(labels, dtype=np.int32))
        hits = agent.update_control(state, action, labels)
        action, agent.adjusted = agent.select_action(new_state)
        state = new_state
        rec['hits'].append(np.array(hits, dtype=np.int32)); rec['action_out'].append(np.array(action, dtype=np.int32))
        rec['adjusted'].append(int(agent.adjusted)); rec['margins'].append(np.array(agent.margins, dtype=np.int32))
        rec['security'].append(np.array(agent.security_factors, dtype=np.int32))
        rec['set_size'].append([h.algorithm.get_set_size() if h.algorithm.counter else 0 for h in agent.learners])
        rec['reward'].append(reward); rec['violation'].append(int(info['total_violations']))
        if i % 10 == 9 or i == steps - 1:
            rec['acc'].append(agent.accuracies.copy())
    rec['final_state'] = np.array(state, dtype=np.float32)
    kind, val = tape.arrays()
    out = {k: np.asarray(v) for k, v in rec.items()}
    out.update(tape_kind=kind, tape_val=val, init_action=np.asarray(init_action, dtype=np.int32),
               init_sec=np.asarray(init_sec, dtype=np.int32), scenario=np.int32(scenario), seed=np.int64(seed),
               a_range=np.asarray(a_range))
    for s, h in enumerate(agent.learners):
        out['coeff%d' % s] = np.asarray(h.algorithm.sv.coeff, dtype=np.float64)
        out['landmarks%d' % s] = np.atleast_2d(h.algorithm.sv.landmarks)
    return out


def _g10(rh, tape, save):
    for scenario, steps in ((0, 120), (2, 150)):
        out = _run_agent(rh, tape, scenario, seed=3, steps=steps, a_range=[0.99, 0.999])
        save('g10_kbrl_s%d' % scenario, **out)


def _g11(rh, tape, save):
    """experiments_kbrl.Evaluator.evaluate plumbing (experiments_kbrl.py:45-55): result dict schema"""
    import scenario_creator as sc
    np.random.seed(11)
    tape.on = False
    rng = np.random.default_rng(0)
    env = sc.create_env(rng, 0)
    agent = sc.create_kbrl_agent(rng, 0, accuracy_range=[0.97, 0.99])
    import io
    import contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        res = agent.run(env, 60)
    tape.on = True
    out = {}
    for k, v in res.items():
        out['key_' + k] = np.asarray(v)
    save('g11_results_schema', **out)


def _kinv_digest(kinv, seed=14):
    """what the fixtures keep of a large Kinv: every 16th row, the diagonal and Kinv @ four fixed probe vectors (a
    700 x 700 float64 matrix would be 4 MB per fixture)"""
    kinv = np.atleast_2d(np.asarray(kinv, dtype=np.float64))
    m = kinv.shape[0]
    probes = np.random.default_rng(seed).standard_normal((m, 4))
    return dict(rows=kinv[::16].copy(), diag=np.diag(kinv).copy(), probes=probes, kp=kinv @ probes)


def _g14(rh, tape, save):
    """G14: Projectron teacher-forced far past the sizes of G9 (m ~ 800): half of the samples revisit an earlier
    state (exactly, or with a small perturbation) with a new action, as KBRL's sample augmentation does, so that
    projections keep happening against a dictionary of several hundred landmarks"""
    from algorithms.kernel import GaussianKernel
    from algorithms.projectron import SVvariable, Projectron
    spread, noise, n, d = 1.3, 0.12, 7000, 11
    rng = np.random.default_rng(140)
    np.random.seed(14)
    tape.clear()
    alg = Projectron(GaussianKernel(SVvariable(), 1))
    xs, ys, fs, ypred, branch, delta, ms, states = [], [], [], [], [], [], [], []
    for i in range(n):
        if states and rng.random() < 0.5:
            s = states[rng.integers(len(states))]
            if rng.random() < 0.5:
                s = (s + rng.normal(0, 0.02, d - 1)).astype(np.float32)
        else:
            s = (rng.random(d - 1) * spread).astype(np.float32)
        states.append(s)
        x = np.append(s, rng.integers(0, 201) / 200)
        score = x[:-1].mean() * 0.8 / spread + 0.35 - x[-1]
        y = -1 if score + rng.normal(0, noise) > 0 else 1
        yp = alg.predict(x)
        f = float(alg.f)
        m0 = alg.counter
        dl = np.nan
        if alg.f * y <= 0:  # projectron.py:41-44 on the reference's arrays
            d_star = alg.Kinv @ alg.K_f
            if np.ndim(d_star) == 0:
                d_star = np.array([d_star], dtype=np.float32)
            dl = float(max(alg.kernel.k_eval(x, x) - d_star @ alg.K_f, 0))
        alg.update(x, y)
        br = 0 if not (f * y <= 0) else (2 if alg.counter > m0 else 1)
        xs.append(x); ys.append(y); fs.append(f); ypred.append(int(yp)); branch.append(br)
        delta.append(dl); ms.append(alg.counter)
    kind, val = tape.arrays()
    dg = _kinv_digest(alg.Kinv)
    save('g14_projectron_long', x=np.asarray(xs), y=np.asarray(ys, dtype=np.int8), f=np.asarray(fs),
         ypred=np.asarray(ypred, dtype=np.int8), branch=np.asarray(branch, dtype=np.int8), delta=np.asarray(delta),
         m=np.asarray(ms, dtype=np.int32), ties=val, landmarks=np.atleast_2d(alg.sv.landmarks),
         coeff=np.asarray(alg.sv.coeff, dtype=np.float64), kinv_rows=dg['rows'], kinv_diag=dg['diag'],
         kinv_probes=dg['probes'], kinv_kp=dg['kp'])


def _g17(rh, tape, save):
    """G17 (round 4): Projectron teacher-forced by the REFERENCE to more than 3,000 landmarks -- where the 50,400-step runs of
    experiments_kbrl.py end (1,431-2,595 landmarks in the reference's own runs on the build's traces, 3,651 in the
    device's).  The stream is tests/test_gpu_kbrl.py::_fast_growing_samples(12400, spread=3.0, revisit=0.35): states spread
    over [0, 3]^10, a third of the samples revisiting an earlier state with a new action.  Kept compact: the states as
    float32, the last coordinate as its grid index; of Kinv (72 MB) the diagonal, every 256th row and Kinv @ four probes."""
    from algorithms.kernel import GaussianKernel
    from algorithms.projectron import SVvariable, Projectron
    spread, revisit, n, d = 3.0, 0.35, 12400, 11
    rng = np.random.default_rng(141)
    np.random.seed(17)
    tape.clear()
    alg = Projectron(GaussianKernel(SVvariable(), 1))
    S, A, ys, fs, ypred, branch, delta, ms, states = [], [], [], [], [], [], [], [], []
    for i in range(n):
        if states and rng.random() < revisit:
            s = states[rng.integers(len(states))]
        else:
            s = (rng.random(10) * spread).astype(np.float32)
        states.append(s)
        a = int(rng.integers(0, 201))
        x = np.append(s, a / 200)
        score = x[:-1].mean() * 0.5 * 1.6 / spread + 0.35 - x[-1]
        y = -1 if score + rng.normal(0, 0.15) > 0 else 1
        yp = alg.predict(x)
        f = float(alg.f)
        m0 = alg.counter
        dl = np.nan
        if alg.f * y <= 0:  # projectron.py:41-44 on the reference's arrays
            d_star = alg.Kinv @ alg.K_f
            if np.ndim(d_star) == 0:
                d_star = np.array([d_star], dtype=np.float32)
            dl = float(max(alg.kernel.k_eval(x, x) - d_star @ alg.K_f, 0))
        alg.update(x, y)
        br = 0 if not (f * y <= 0) else (2 if alg.counter > m0 else 1)
        S.append(s); A.append(a); ys.append(y); fs.append(f); ypred.append(int(yp)); branch.append(br)
        delta.append(dl); ms.append(alg.counter)
    kind, val = tape.arrays()
    kinv = np.atleast_2d(np.asarray(alg.Kinv, dtype=np.float64))
    probes = np.random.default_rng(17).standard_normal((kinv.shape[0], 4))
    save('g17_projectron_3000', state=np.asarray(S, dtype=np.float32), a=np.asarray(A, dtype=np.uint8),
         y=np.asarray(ys, dtype=np.int8), f=np.asarray(fs), ypred=np.asarray(ypred, dtype=np.int8),
         branch=np.asarray(branch, dtype=np.int8), delta=np.asarray(delta), m=np.asarray(ms, dtype=np.int32), ties=val,
         coeff=np.asarray(alg.sv.coeff, dtype=np.float64), kinv_rows=kinv[::256].copy(), kinv_diag=np.diag(kinv).copy(),
         kinv_probes=probes, kinv_kp=kinv @ probes)


def _long_control(rh, tape, save, name, profile, seed):
    import tempfile
    from ranslice.fading import synth_traces
    rh.setup(tempfile.mkdtemp(prefix='refwork_%s_' % name), synth_traces(10000, profile))
    out = _run_agent(rh, tape, 0, seed=seed, steps=2200, a_range=[0.99, 0.999])
    # only the tie-break draws of the agent are needed by the teacher-forced replay (the simulator is not replayed)
    ties = out['tape_val'][out['tape_kind'] == 6]
    out = {k: v for k, v in out.items() if k not in ('tape_kind', 'tape_val', 'reward')}
    out['ties'] = ties
    out['acc'] = out['acc'][-1:]          # the accuracy tables of the last step only
    for k in ('action_in', 'labels', 'hits', 'action_out', 'margins', 'security', 'set_size', 'violation', 'adjusted'):
        out[k] = out[k].astype(np.int16)
    out['profile'] = np.array(profile)
    save(name, **out)


def _g15(rh, tape, save):
    """G15: KBRL_Control teacher-forced over 2,200 steps of scenario_0 on the 10,000-column traces of the first
    synthetic profile (deep, flat fades: the regime in which dictionaries grow to several hundred landmarks)"""
    import os
    seed = int(os.environ.get('G15_SEED', '6'))   # the committed fixture: the first seed whose run has no f == 0 tie
    _long_control(rh, tape, save, os.environ.get('G15_NAME', 'g15_kbrl_long_s0'), 'sos', seed)


def _g16(rh, tape, save):
    """G16: the same on the tapped-delay-line traces (ranslice.fading.synth_fading_tdl), the regime closer to the
    paper's: few violations, small dictionaries"""
    _long_control(rh, tape, save, 'g16_kbrl_long_tdl_s0', 'tdl', 5)


def generate(rh, tape, save):
    _g9(rh, tape, save)
    _g10(rh, tape, save)
    _g11(rh, tape, save)


def generate_long(rh, tape, save, which):
    if 'G14' in which:
        _g14(rh, tape, save)
    if 'G15' in which:
        _g15(rh, tape, save)
    if 'G16' in which:
        _g16(rh, tape, save)
    if 'G17' in which:
        _g17(rh, tape, save)
