#!/usr/bin/env python3
"""Developer aid: rewrite rs_embb.hip's section marks so that the PROFILE build (make -C network-slicing_amd/csrc profile) times the
inside of a contested PF round -- block-round passes, trip -- instead of the slot's sections; tools/pf_micro_profile.py reads the
result.  usage: python tools/pf_micro_patch.py apply | restore   (works on the file in place; restore = git checkout)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, 'network-slicing_amd', 'csrc', 'rs_embb.hip')


def apply():
    s = open(P).read()
    def rep(a, b, optional=False):
        nonlocal s
        if optional and a not in s:
            return
        assert a in s, a[:60]
        s = s.replace(a, b)
    rep('''#define SEC_MARK(i)                                              \\
    {                                                            \\
        unsigned long long t_ = __builtin_amdgcn_s_memtime();    \\
        sec_acc[i] += t_ - sec_t0;                               \\
        sec_t0 = t_;                                             \\
    }''', '''#define MM(i)                                                    \\
    {                                                            \\
        unsigned long long t_ = __builtin_amdgcn_s_memtime();    \\
        sec_acc[i] += t_ - sec_t0;                               \\
        sec_t0 = t_;                                             \\
    }
#define SEC_MARK(i) MM(12)''')
    rep('''#else
#define SEC_DECL
#define SEC_MARK(i)
#define SEC_FLUSH(buf)
#endif''', '''#else
#define SEC_DECL
#define SEC_MARK(i)
#define MM(i)
#define SEC_FLUSH(buf)
#endif''')
    rep('''                    if (more) stat += 1u << 18;''', '''                    if (more) stat += 1u << 18;
                    MM(0)''')
    rep('''                        const bool cont = bmode && m > 0.0;''', '''                        const bool cont = bmode && m > 0.0;
                        MM(1)''')
    rep('''                            if (cont) e = run1 ? rate_d / tmax1 : 0.0;
                        }
                        SEC_MARK(11)''', '''                            MM(2)
                            if (cont) e = run1 ? rate_d / tmax1 : 0.0;
                        }
                        MM(2)''')
    rep('''                        const int ustar = __ffs((int)group_ballot<G>(e == Lk, gbase)) - 1;''',
        '''                        const int ustar = __ffs((int)group_ballot<G>(e == Lk, gbase)) - 1;
                        MM(4)''')
    rep('''                        SEC_MARK(12)
                        const int T = group_sum<G>(cnt);''', '''                        MM(5)
                        const int T = group_sum<G>(cnt);''')
    rep('''                    bool rmode = false;
                    if (RS_RANK != 0 && G >= 16 && wave_any(more && !bmode && rk_ok)) {''', '''                    MM(6)
                    bool rmode = false;
                    if (RS_RANK != 0 && G >= 16 && wave_any(more && !bmode && rk_ok)) {''', optional=True)
    rep('''                            __builtin_amdgcn_wave_barrier();
                            const bool elane = gl < 16;''', '''                            MM(10)
                            sec_acc[14] += 1;  // rank rounds (wave level; read from the slowest wave's bank only)
                            __builtin_amdgcn_wave_barrier();
                            const bool elane = gl < 16;''', optional=True)
    rep('''                            const int rl = group_min<G>((elane && ev < 0.0) ? rank : 255);''', '''                            MM(11)
                            const int rl = group_min<G>((elane && ev < 0.0) ? rank : 255);''', optional=True)
    rep('''                    const bool tm = more && !bmode && !rmode;  // tasks on a trip this round''', '''                    MM(3)
                    const bool tm = more && !bmode && !rmode;  // tasks on a trip this round''', optional=True)
    rep('''                    const bool tm = more && !bmode;  // tasks on a trip this round''', '''                    MM(6)
                    const bool tm = more && !bmode;  // tasks on a trip this round''', optional=True)
    rep('''                        int take = 0;
                        SEC_MARK(11)
                        if (tm) {''', '''                        int take = 0;
                        MM(7)
                        if (tm) {''')
    rep('''                        SEC_MARK(12)
                        const int tk = bperm(take, gbase + idx);''', '''                        MM(8)
                        const int tk = bperm(take, gbase + idx);''')
    rep('''                            } else {
                                need_full = true;
                            }
                        }
                    }
                }
            }

            SEC_MARK(3)''', '''                            } else {
                                need_full = true;
                            }
                        }
                        MM(9)
                    }
                }
            }

            SEC_MARK(3)''')
    open(P, 'w').write(s)


if __name__ == '__main__':
    if sys.argv[1:] == ['apply']:
        apply()
    else:
        subprocess.check_call(['git', 'checkout', P], cwd=ROOT)
