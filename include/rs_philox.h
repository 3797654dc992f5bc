/* rs_philox.h -- counter-based random streams for the batched RAN-slice simulator.
 *
 * The reference threads ONE sequential numpy Generator through every slice of an env
 * (reference experiments_kbrl.py:46, scenario_creator.py:146-166) plus the unseeded global
 * np.random (traffic_generators.py:66,96,97).  That order is inherently serial.  The
 * build replaces it with Philox4x32-10 streams addressed by
 *     key     = 64-bit replica seed
 *     counter = (draw index, 0, UE serial within the slice [0 = the slice itself], slice id)
 * so that every (replica, slice, UE) draws independently of all others and a wavefront
 * can advance many UEs at once.  The fading walker's redraws (channel_models.py:179-183) are
 * addressed by TIME instead of by draw index,
 *     counter = (absolute slot, 1 + attempt within the slot, UE serial, slice id),
 * so that a UE's fading trajectory -- hence every channel estimate of a step -- is a function of
 * its arrival state alone and can be evaluated ahead of the scheduling loop (rs_walker_redraw).  The distribution of every draw equals the reference's
 * (uniform, exponential, uniform integer, +-1, normal); the bit patterns do not, which is
 * why parity is argued in two hops: reference == oracle on a recorded tape,
 * oracle == HIP on these streams.
 *
 * Compiles as C (oracle) and as HIP device code.  Uses rs_detmath.h for log/sqrt so that
 * both sides produce identical bits.
 */
#ifndef RS_PHILOX_H
#define RS_PHILOX_H

#include "rs_detmath.h"

typedef struct {
    uint32_t key0, key1; /* replica seed */
    uint32_t slice;      /* slice id */
    uint32_t serial;     /* 0 = slice-level stream, >=1 = UE serial */
    uint32_t ctr;        /* next draw index */
} rs_stream;

RS_HD void rs_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                            uint32_t k1, uint32_t* o0, uint32_t* o1) {
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    *o0 = c0;
    *o1 = c1;
}

/* uniform double in [0,1) with 53 random bits; advances the stream by one */
RS_HD double rs_stream_uniform(rs_stream* s) {
    uint32_t a, b;
    rs_philox4x32_10(s->ctr, 0u, s->serial, s->slice, s->key0, s->key1, &a, &b);
    s->ctr += 1u;
    uint64_t x = (((uint64_t)a << 32) | (uint64_t)b) >> 11;
    return (double)x * 1.1102230246251565e-16; /* 2^-53 */
}

/* exponential with the given scale (mean): -log(1-u)*scale, u in [0,1) */
RS_HD double rs_stream_exponential(rs_stream* s, double scale) {
    double u = rs_stream_uniform(s);
    return (-RS_LOG_CALL(1.0 - u)) * scale;
}

/* uniform integer in [0,n) */
RS_HD int64_t rs_stream_integers(rs_stream* s, int64_t n) {
    double u = rs_stream_uniform(s);
    int64_t v = (int64_t)(u * (double)n);
    return v < n ? v : n - 1;
}

/* -1 or +1 with probability 1/2 */
RS_HD int rs_stream_pm1(rs_stream* s) { return rs_stream_uniform(s) < 0.5 ? -1 : 1; }

/* SINRSelectiveFading.get_snr leaving [0, T) (channel_models.py:179-183): new index uniform in [0, T) and a new
 * direction, both from ONE Philox block addressed by (absolute slot, attempt) -- stateless, see the header. */
RS_HD void rs_walker_redraw(uint32_t key0, uint32_t key1, uint32_t slice, uint32_t serial, uint32_t now,
                            uint32_t attempt, int T, int* findex, int* fstep) {
    uint32_t a, b;
    rs_philox4x32_10(now, 1u + attempt, serial, slice, key0, key1, &a, &b);
    *findex = (int)(((uint64_t)a * (uint64_t)(uint32_t)T) >> 32);
    *fstep = (b >> 31) ? 1 : -1;
}

/* normal(loc, scale): Marsaglia polar method (log, sqrt, / only -> deterministic) */
RS_HD double rs_stream_normal(rs_stream* s, double loc, double scale) {
    double v1, v2, r2;
    do {
        v1 = 2.0 * rs_stream_uniform(s) - 1.0;
        v2 = 2.0 * rs_stream_uniform(s) - 1.0;
        r2 = v1 * v1 + v2 * v2;
    } while (r2 >= 1.0 || r2 == 0.0);
    double z = v1 * RS_SQRT((-2.0 * RS_LOG_CALL(r2)) / r2);
    return loc + scale * z;
}

#endif /* RS_PHILOX_H */
