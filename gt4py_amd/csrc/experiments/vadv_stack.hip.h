// vertical_advection_dycore with the top of every column kept on chip between its two sweeps.
//
// The stencil is /root/reference/tests/cartesian_tests/integration_tests/multi_feature_tests/stencil_definitions.py:235-313:
// a FORWARD sweep that builds the Thomas coefficients ccol / dcol level by level (each level reads wcon at two I
// offsets and two levels, u_stage at three levels, u_pos, utens, utens_stage) and a BACKWARD substitution that reads
// ccol, dcol and u_pos again and overwrites utens_stage.  ccol and dcol are temporaries: nothing outside the kernel sees
// them, so the levels that fit on chip never go to memory at all -- RL levels in registers, LL levels in LDS, one
// wave per workgroup and one workgroup per SIMD exactly as tridiag_pipe_kernel (tridiag_stack.hip.h) does for its
// in/out fields; only the first dK - RL - LL levels of a column spill to the scratch fields `ccol` / `dcol`.
//
// Algorithmic bytes: 5 fields read + 1 written = 48 B per point; unavoidable extra: u_pos is read by both sweeps
// (+8 B) and the spilled levels cost 32 B each (2 fields x write + read).
//
// Every expression keeps the operation order of the statement it restates (bit-exact against the oracle): the shared
// sum wcon[1,0,k] + wcon[0,0,k] is formed once per level and used by level k (gav) and level k-1 (gcv), which is the
// same double either way.
#pragma once

#include "common.hip.h"

namespace gt4mi {

struct VadvFields {
    View<const double> wcon, u_stage, u_pos, utens;
    View<double> utens_stage;
    View<double> ccol, dcol;  // scratch for the spilled levels (levels [0, dK - RL - LL))
};

template <int U>
struct VadvFwdBatch {
    double w0[U], w1[U], us[U];  // wcon[0,0,.], wcon[1,0,.], u_stage at levels k+1 .. k+U
    double up[U], ut[U], ts[U];  // u_pos, utens, utens_stage at levels k .. k+U-1
};
template <int U>
struct VadvBwdBatch {
    double c[U], d[U], up[U];
};

// one thread per column, plain loops, every temporary level through memory: the shape of the generated kernel without
// any cache -- the bit-exact yardstick of the micro-benchmark (and of the tests of the kernel below)
__global__ void __launch_bounds__(256) vadv_plain_kernel(VadvFields f, double dtr, double bet_m, double bet_p, int dI, int dJ, int dK) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    const int j = blockIdx.y * 4 + threadIdx.y;
    if (i >= dI || j >= dJ) return;
    const double* w = f.wcon.p + (int64_t)j * f.wcon.sj + i;
    const double* us = f.u_stage.p + (int64_t)j * f.u_stage.sj + i;
    const double* up = f.u_pos.p + (int64_t)j * f.u_pos.sj + i;
    const double* ut = f.utens.p + (int64_t)j * f.utens.sj + i;
    double* ts = f.utens_stage.p + (int64_t)j * f.utens_stage.sj + i;
    double* cc = f.ccol.p + (int64_t)j * f.ccol.sj + i;
    double* dc = f.dcol.p + (int64_t)j * f.dcol.sj + i;
    double cp = 0, dp = 0;
    for (int k = 0; k < dK; ++k) {
        double ccol = 0, dcol;
        if (k == 0) {
            const double gcv = 0.25 * (w[(int64_t)(k + 1) * f.wcon.sk + 1] + w[(int64_t)(k + 1) * f.wcon.sk]);
            const double cs = gcv * bet_m;
            ccol = gcv * bet_p;
            const double bcol = dtr - ccol;
            const double corr = (-cs) * (us[(int64_t)(k + 1) * f.u_stage.sk] - us[(int64_t)k * f.u_stage.sk]);
            dcol = (((dtr * up[(int64_t)k * f.u_pos.sk]) + ut[(int64_t)k * f.utens.sk]) + ts[(int64_t)k * f.utens_stage.sk]) + corr;
            const double divided = 1.0 / bcol;
            ccol = ccol * divided;
            dcol = dcol * divided;
        } else {
            const double gav = (-0.25) * (w[(int64_t)k * f.wcon.sk + 1] + w[(int64_t)k * f.wcon.sk]);
            const double as_ = gav * bet_m;
            const double acol = gav * bet_p;
            double bcol, corr;
            if (k < dK - 1) {
                const double gcv = 0.25 * (w[(int64_t)(k + 1) * f.wcon.sk + 1] + w[(int64_t)(k + 1) * f.wcon.sk]);
                const double cs = gcv * bet_m;
                ccol = gcv * bet_p;
                bcol = (dtr - acol) - ccol;
                corr = ((-as_) * (us[(int64_t)(k - 1) * f.u_stage.sk] - us[(int64_t)k * f.u_stage.sk])) -
                       (cs * (us[(int64_t)(k + 1) * f.u_stage.sk] - us[(int64_t)k * f.u_stage.sk]));
            } else {
                bcol = dtr - acol;
                corr = (-as_) * (us[(int64_t)(k - 1) * f.u_stage.sk] - us[(int64_t)k * f.u_stage.sk]);
            }
            dcol = (((dtr * up[(int64_t)k * f.u_pos.sk]) + ut[(int64_t)k * f.utens.sk]) + ts[(int64_t)k * f.utens_stage.sk]) + corr;
            const double divided = 1.0 / (bcol - (cp * acol));
            ccol = ccol * divided;
            dcol = (dcol - (dp * acol)) * divided;
        }
        cc[(int64_t)k * f.ccol.sk] = ccol;
        dc[(int64_t)k * f.dcol.sk] = dcol;
        cp = ccol;
        dp = dcol;
    }
    double data = dp;
    ts[(int64_t)(dK - 1) * f.utens_stage.sk] = dtr * (data - up[(int64_t)(dK - 1) * f.u_pos.sk]);
    for (int k = dK - 2; k >= 0; --k) {
        data = dc[(int64_t)k * f.dcol.sk] - (cc[(int64_t)k * f.ccol.sk] * data);
        ts[(int64_t)k * f.utens_stage.sk] = dtr * (data - up[(int64_t)k * f.u_pos.sk]);
    }
}

// A wave-uniform pointer the optimiser knows nothing about: addresses derived from it cannot be merged with equal
// addresses computed elsewhere.  Without it the compiler keeps the per-level addresses of u_pos / utens_stage that the
// forward sweep formed alive in (accumulator) registers until the backward sweep uses them again -- 4 registers per
// on-chip level, which is exactly the space the cache is supposed to get.
// (An offset is laundered, not the pointer: the pointer keeps its global address space.)
template <typename P>
__device__ __forceinline__ P* scalar_launder(P* p) {
    int64_t zero = 0;
    asm volatile("" : "+s"(zero));
    return p + zero;
}
// Pins values to this point of the instruction stream.  A level whose results only go to registers has no side
// effect, and instruction selection is free to sink its arithmetic to the first use of the result -- the other sweep --
// across every __builtin_amdgcn_sched_barrier in between (those order the scheduler, not the selector): all loads
// of the on-chip levels would then stay live (6 doubles per level) until the backward sweep starts.
__device__ __forceinline__ void pin_here(double a, double b) { asm volatile("" ::"v"(a), "v"(b)); }

// RL register levels + LL LDS levels at the top of the column, batches of U levels, loads of batch n+1 in flight while
// batch n is computed.  Needs dK - RL - LL >= 1 (level 0 goes through memory; the host picks a smaller variant
// otherwise) and unit I stride.  Every access is <running scalar row pointer>[lane]: the per-level address arithmetic
// is two scalar adds per field and no address lives in a vector register.
// PIPE = false waits for every batch of loads right after issuing it (no load is in flight during arithmetic: what a
// loop that loads, then computes, does); SADDR = false folds the lane into the row pointers (one 64-bit vector address
// per access instead of scalar base + lane) -- both only for the ablation in the micro-benchmark.
// SLOTS > 0 (round 5 experiment): the levels that spill do not go to the column's own place in full-size ccol / dcol arrays but to
// slot (blockIdx.x % SLOTS) of a small area that is reused block after block -- does a spill area of a few MB stay in the caches?
// (A probe: two live blocks of one slot would collide; the micro-benchmark checks the result.)
// NTM (round 5 probe): 1 = nontemporal loads the way the code generator's mode 5 places them (u_stage, utens, utens_stage in the forward
// sweep, u_pos at its last use in the backward sweep; wcon -- read at i and i + 1 -- and u_pos's first read stay cacheable);
// 2 = 1 + wcon loaded ONCE, nontemporally, the i + 1 value taken from the next lane (lane 63 loads its own): does a vertical advection
// whose only cacheable stream is u_pos keep more of it on chip between the sweeps?
template <int RL, int LL, int U, bool PIPE = true, bool SADDR = true, int SLOTS = 0, int NTM = 0>
__global__ void __launch_bounds__(64)
vadv_pipe_kernel(VadvFields f, double dtr, double bet_m, double bet_p, int dI, int dJ, int dK, unsigned tiles_i) {
    static_assert(RL % U == 0 && LL % U == 0 && RL >= U, "level ranges are processed in batches of U");
    static_assert((LL + RL) / U >= 2, "the top batch (which must not read level dK) is loaded by the on-chip loop");
    __shared__ double lds[LL > 0 ? LL * 2 * 64 : 1];
    const unsigned bi = blockIdx.x % tiles_i;
    const unsigned j = blockIdx.x / tiles_i;
    const unsigned lane_id = threadIdx.x;
    if ((int)(bi * 64 + lane_id) >= dI) return;
    const unsigned lane = SADDR ? lane_id : 0u;
    const int64_t ib = (int64_t)bi * 64 + (SADDR ? 0u : lane_id);
    auto issued = [&]() {
        if (!PIPE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    const int64_t w_sk = f.wcon.sk, us_sk = f.u_stage.sk, up_sk = f.u_pos.sk, ut_sk = f.utens.sk, ts_sk = f.utens_stage.sk,
                  cc_sk = SLOTS > 0 ? (int64_t)SLOTS * 64 : f.ccol.sk, dc_sk = SLOTS > 0 ? (int64_t)SLOTS * 64 : f.dcol.sk;
    const int64_t spill0 = SLOTS > 0 ? (int64_t)(blockIdx.x % (SLOTS > 0 ? SLOTS : 1)) * 64 + (SADDR ? 0u : lane_id) : 0;

    // forward sweep: load cursors (wcon / u_stage run one level ahead of the others) and spill cursors
    const double* qw = scalar_launder(f.wcon.p + (int64_t)j * f.wcon.sj + ib + w_sk);
    const double* qus = scalar_launder(f.u_stage.p + (int64_t)j * f.u_stage.sj + ib);
    const double* qup = scalar_launder(f.u_pos.p + (int64_t)j * f.u_pos.sj + ib);
    const double* qut = scalar_launder(f.utens.p + (int64_t)j * f.utens.sj + ib);
    const double* qts = scalar_launder((const double*)f.utens_stage.p + (int64_t)j * f.utens_stage.sj + ib);
    double* qcc = scalar_launder(SLOTS > 0 ? f.ccol.p + spill0 : f.ccol.p + (int64_t)j * f.ccol.sj + ib);
    double* qdc = scalar_launder(SLOTS > 0 ? f.dcol.p + spill0 : f.dcol.p + (int64_t)j * f.dcol.sj + ib);

    const int A = dK - LL - RL;  // levels [0, A) spill to memory
    double cp, dp;               // ccol[k-1], dcol[k-1]
    double ws;                   // wcon[1,0,k] + wcon[0,0,k] of the level about to be computed
    double usm, usc;             // u_stage[k-1], u_stage[k]
    double C[RL], D[RL];         // the last RL levels; compile-time indices only

    using FB = VadvFwdBatch<U>;
    using BB = VadvBwdBatch<U>;
    // one interior level: consumes the carried values and the level's own loads, leaves ccol/dcol of the level in cp/dp
    auto east = [&](double w0n, double w1n) {  // wcon[i + 1]: NTM 2 takes it from the next lane (lane 63 loaded its own)
        if constexpr (NTM == 2) {
            const double from_next = lane_shift<double, false>(w0n);
            return lane_id == 63 ? w1n : from_next;
        } else {
            return w1n;
        }
    };
    auto mid = [&](double w0n, double w1n_in, double usn, double up, double ut, double ts) {
        const double w1n = east(w0n, w1n_in);
        const double wsn = w1n + w0n;
        const double gav = (-0.25) * ws;
        const double gcv = 0.25 * wsn;
        const double as_ = gav * bet_m;
        const double cs = gcv * bet_m;
        const double acol = gav * bet_p;
        double ccol = gcv * bet_p;
        const double bcol = (dtr - acol) - ccol;
        const double corr = ((-as_) * (usm - usc)) - (cs * (usn - usc));
        double dcol = (((dtr * up) + ut) + ts) + corr;
        const double divided = 1.0 / (bcol - (cp * acol));
        ccol = ccol * divided;
        dcol = (dcol - (dp * acol)) * divided;
        cp = ccol;
        dp = dcol;
        ws = wsn;
        usm = usc;
        usc = usn;
    };
    // one level of loads at the cursors: wcon / u_stage of the level above, the rest of the level itself
    auto load_level = [&](double& w0, double& w1, double& us, double& up, double& ut, double& ts, bool above = true) {
        if (above) {
            if constexpr (NTM == 2) {
                w0 = __builtin_nontemporal_load(qw + lane);
                w1 = 0.0;  // (taken from the next lane where the value is USED -- a shift here would wait for the load)
                if (lane_id == 63) w1 = __builtin_nontemporal_load(qw + lane + 1);
            } else {
                w0 = qw[lane];
                w1 = qw[lane + 1];
            }
            us = NTM ? __builtin_nontemporal_load(qus + lane) : qus[lane];
        }
        up = qup[lane];
        ut = NTM ? __builtin_nontemporal_load(qut + lane) : qut[lane];
        ts = NTM ? __builtin_nontemporal_load(qts + lane) : qts[lane];
        qw += w_sk;
        qus += us_sk;
        qup += up_sk;
        qut += ut_sk;
        qts += ts_sk;
    };
    auto load = [&](FB& b, bool top = false) {
#pragma unroll
        for (int u = 0; u < U; ++u) load_level(b.w0[u], b.w1[u], b.us[u], b.up[u], b.ut[u], b.ts[u], !(top && u == U - 1));
    };
    auto spill = [&]() {
        qcc[lane] = cp;  // read again by the backward sweep: keep cacheable
        qdc[lane] = dp;
        qcc += cc_sk;
        qdc += dc_sk;
    };
    auto forward_mem = [&](const FB& b) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            mid(b.w0[u], b.w1[u], b.us[u], b.up[u], b.ut[u], b.ts[u]);
            spill();
        }
    };

    // ---- FORWARD ----------------------------------------------------------------------------------
    FB B[2];
    const int head = (A - 1) % U;  // levels [1, 1 + head) come first so that whole batches end exactly at A
    {
        // level 0 and the head levels are loaded together
        const double us0 = qus[lane];
        qus += us_sk;
        double w0a, w1a, us1, up0, ut0, ts0;
        load_level(w0a, w1a, us1, up0, ut0, ts0);
        FB h;
#pragma unroll
        for (int u = 0; u < U - 1; ++u)
            if (u < head) load_level(h.w0[u], h.w1[u], h.us[u], h.up[u], h.ut[u], h.ts[u]);  // wave-uniform
        load(B[0]);  // the first whole batch
        issued();
        {  // interval(0, 1)
            const double wsn = east(w0a, w1a) + w0a;
            const double gcv = 0.25 * wsn;
            const double cs = gcv * bet_m;
            double ccol = gcv * bet_p;
            const double bcol = dtr - ccol;
            const double corr = (-cs) * (us1 - us0);
            double dcol = (((dtr * up0) + ut0) + ts0) + corr;
            const double divided = 1.0 / bcol;
            ccol = ccol * divided;
            dcol = dcol * divided;
            cp = ccol;
            dp = dcol;
            ws = wsn;
            usm = us0;
            usc = us1;
            spill();
        }
#pragma unroll
        for (int u = 0; u < U - 1; ++u)
            if (u < head) {
                mid(h.w0[u], h.w1[u], h.us[u], h.up[u], h.ut[u], h.ts[u]);
                spill();
            }
    }
    int k = 1 + head;
    // invariant: B[0] holds (or is receiving) the batch that starts at level k
    while (k + 2 * U <= A) {
        load(B[1]);
        issued();
        forward_mem(B[0]);
        load(B[0]);  // may already be the first on-chip batch: same form, levels are contiguous
        issued();
        forward_mem(B[1]);
        k += 2 * U;
    }
    if (k + U <= A) {  // an odd number of memory batches
        load(B[1]);
        issued();
        forward_mem(B[0]);
        B[0] = B[1];
        k += U;
    }
    // k == A: the LL + RL on-chip levels, fully unrolled, B[b & 1] is a compile-time choice
    constexpr int NBC = (LL + RL) / U;
#pragma unroll
    for (int b = 0; b < NBC; ++b) {
        __builtin_amdgcn_sched_barrier(0);  // keeps the loads of later batches out of this one (registers)
        if (b + 1 < NBC) load(B[(b + 1) & 1], b + 2 == NBC);
        issued();
        const FB& c = B[b & 1];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int l = b * U + u;  // level A + l
            if (l == LL + RL - 1) {   // interval(-1, None): no gcv / ccol
                const double gav = (-0.25) * ws;
                const double as_ = gav * bet_m;
                const double acol = gav * bet_p;
                const double bcol = dtr - acol;
                const double corr = (-as_) * (usm - usc);
                double dcol = (((dtr * c.up[u]) + c.ut[u]) + c.ts[u]) + corr;
                const double divided = 1.0 / (bcol - (cp * acol));
                dcol = (dcol - (dp * acol)) * divided;
                dp = dcol;
            } else {
                mid(c.w0[u], c.w1[u], c.us[u], c.up[u], c.ut[u], c.ts[u]);
            }
            if (l < LL) {
                lds[(l * 2 + 0) * 64 + lane_id] = cp;
                lds[(l * 2 + 1) * 64 + lane_id] = dp;
            } else {
                C[l - LL] = cp;
                D[l - LL] = dp;
                pin_here(cp, dp);
            }
        }
    }

    // ---- BACKWARD: fresh cursors, top of the column downwards ---------------------------------------
    const double* rup = scalar_launder(f.u_pos.p + (int64_t)j * f.u_pos.sj + ib + (int64_t)(dK - 1) * up_sk);
    double* rts = scalar_launder(f.utens_stage.p + (int64_t)j * f.utens_stage.sj + ib + (int64_t)(dK - 1) * ts_sk);
    const double* rcc = scalar_launder((const double*)f.ccol.p + (SLOTS > 0 ? spill0 : (int64_t)j * f.ccol.sj + ib) + (int64_t)(A - 1) * cc_sk);
    const double* rdc = scalar_launder((const double*)f.dcol.p + (SLOTS > 0 ? spill0 : (int64_t)j * f.dcol.sj + ib) + (int64_t)(A - 1) * dc_sk);
    // the spilled levels' own u_pos cursor: their loads are issued before the on-chip levels' are finished
    const double* rupm = scalar_launder(f.u_pos.p + (int64_t)j * f.u_pos.sj + ib + (int64_t)(A - 1) * up_sk);
    auto loadb = [&](BB& b) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            b.c[u] = rcc[lane];
            b.d[u] = rdc[lane];
            b.up[u] = NTM ? __builtin_nontemporal_load(rupm + lane) : rupm[lane];
            rcc -= cc_sk;
            rdc -= dc_sk;
            rupm -= up_sk;
        }
    };
    double data;
    auto emit = [&](double up) {
        __builtin_nontemporal_store(dtr * (data - up), rts + lane);
        rts -= ts_sk;
    };
    auto backward_mem = [&](const BB& b) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            data = b.d[u] - (b.c[u] * data);
            emit(b.up[u]);
        }
    };
    // u_pos of the on-chip levels: three batches of U in rotation, two in flight ahead of the one being used
    double P[3][U];
    auto loadp = [&](double (&p)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            p[u] = NTM ? __builtin_nontemporal_load(rup + lane) : rup[lane];
            rup -= up_sk;
        }
    };
    loadp(P[0]);
    loadp(P[1]);
    BB Cb[2];
    int kb = A - 1;
    if (kb - U + 1 >= 0) loadb(Cb[0]);  // in flight during the on-chip part of the sweep
    issued();
#pragma unroll
    for (int b = 0; b < NBC; ++b) {
        __builtin_amdgcn_sched_barrier(0);
        if (b + 2 < NBC) loadp(P[(b + 2) % 3]);
        issued();
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int l = LL + RL - 1 - (b * U + u);  // level A + l
            if (l == LL + RL - 1) {
                data = D[RL - 1];
            } else if (l >= LL) {
                data = D[l - LL] - (C[l - LL] * data);
            } else {
                const double c = lds[(l * 2 + 0) * 64 + lane_id], d = lds[(l * 2 + 1) * 64 + lane_id];
                data = d - (c * data);
            }
            emit(P[b % 3][u]);
        }
    }
    // invariant: Cb[0] holds the batch kb, kb - 1, ... whenever a whole batch is left
    while (kb - 2 * U + 1 >= 0) {
        loadb(Cb[1]);
        issued();
        backward_mem(Cb[0]);
        if (kb - 3 * U + 1 >= 0) loadb(Cb[0]);
        issued();
        backward_mem(Cb[1]);
        kb -= 2 * U;
    }
    if (kb - U + 1 >= 0) {
        backward_mem(Cb[0]);
        kb -= U;
    }
    {  // what is left at the bottom (fewer than U levels), loaded together
        BB t;
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (u <= kb) {
                t.c[u] = rcc[lane];
                t.d[u] = rdc[lane];
                t.up[u] = NTM ? __builtin_nontemporal_load(rupm + lane) : rupm[lane];
                rcc -= cc_sk;
                rdc -= dc_sk;
                rupm -= up_sk;
            }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (u <= kb) {
                data = t.d[u] - (t.c[u] * data);
                emit(t.up[u]);
            }
    }
}

}  // namespace gt4mi
