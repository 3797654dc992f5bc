#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (kernel trace, optionally PMC counters) as text.
usage: python tools/rocpd_summary.py <results.db> [--last N] [> profiles/<name>.txt]
--steps K --anchor <kernel name prefix>: per-kernel ms per step over the last K steps (a step = one launch of the anchor).
--last N also prints the mean duration of the last N launches of the dominant kernel (= a bench run's timed
region, which is what bench.py's HIP-event kernel time covers)."""
import sqlite3
import sys


def main(path, last=0, steps=0, anchor=None):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                     "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print('# kernel trace summary of %s' % path)
    print('%-60s %8s %14s %12s %12s %12s %7s' % ('kernel', 'calls', 'total_ns', 'avg_ns', 'min_ns', 'max_ns', 'pct'))
    for n, k, t, a, mn, mx in rows:
        print('%-60s %8d %14d %12.0f %12d %12d %6.2f%%' % (n[:60], k, t, a, mn, mx, 100.0 * t / tot))
    if last and rows:
        top = rows[0][0]
        d = [r[0] for r in c.execute("select end - start from kernels where name = ? order by start desc limit ?", (top, last))]
        print('\n# last %d launches of %s: mean %.0f ns (min %d, max %d)' % (len(d), top[:60], sum(d) / max(1, len(d)), min(d), max(d)))
    if steps and anchor:
        # per-step account of the last `steps` steps: every launch from the start of the steps-th last launch of the anchor kernel
        st = [r[0] for r in c.execute("select start from kernels where name like ? order by start desc limit ?", (anchor + '%', steps))]
        if len(st) == steps:
            t0 = st[-1]
            acc = c.execute("select name, count(*), sum(end-start) from kernels where start >= ? group by name order by 3 desc", (t0,)).fetchall()
            span = c.execute("select max(end) - ? from kernels", (t0,)).fetchone()[0]
            print('\n# the last %d steps (from the %d-th last launch of %s on): wall %.3f ms per step; per kernel: launches per step, ms per step' % (steps, steps, anchor, span / steps / 1e6))
            for n, k, t in acc:
                print('   %-60s %7.2f %9.4f' % (n[:60], k / steps, t / steps / 1e6))
            print('   %-60s %7s %9.4f' % ('(sum of kernel time)', '', sum(a[2] for a in acc) / steps / 1e6))
    extra = [x for x in ('vgpr_count', 'accum_vgpr_count', 'sgpr_count', 'lds_size', 'scratch_size', 'workgroup_size', 'grid_size') if x in cols]
    if extra:
        print('\n# per-kernel resources (%s)' % ', '.join(extra))
        for r in c.execute("select distinct name, %s from kernels" % ', '.join(extra)):
            print('  ', r)
    try:
        pm = c.execute("select kernel_name, counter_name, avg(v), sum(v), count(*) from (select kernel_name, "
                       "counter_name, dispatch_id, sum(value) as v from counters_collection group by 1, 2, 3) "
                       "group by 1, 2").fetchall()
    except sqlite3.Error as e:
        pm = []
        print('\n# no PMC data (%s)' % e)
    if pm:
        print('\n# PMC counters: kernel, counter, mean per dispatch, sum, dispatches')
        for r in pm:
            print('   %-48s %-28s %16.1f %18.1f %6d' % (r[0][:48], r[1], r[2], r[3], r[4]))
        if last:
            # the population is still growing during the burn-in: what belongs beside the bench line is the
            # mean over the LAST dispatches of each kernel (the timed region)
            print('\n# PMC counters, mean over the last %d dispatches of each kernel' % last)
            names = sorted({(r[0], r[1]) for r in pm})
            for kn, cn in names:
                v = [x[0] for x in c.execute(
                    "select sum(value) from counters_collection where kernel_name = ? and counter_name = ? "
                    "group by dispatch_id order by dispatch_id desc limit ?", (kn, cn, last))]
                if v:
                    print('   %-48s %-28s %16.1f   (last %d)' % (kn[:48], cn, sum(v) / len(v), len(v)))


if __name__ == '__main__':
    def opt(name, default=None):
        return sys.argv[sys.argv.index(name) + 1] if name in sys.argv else default
    main(sys.argv[1], int(opt('--last', 0)), int(opt('--steps', 0)), opt('--anchor'))
