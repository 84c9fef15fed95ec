#!/usr/bin/env python3
"""would a k-way partitioned Riccati solve (k = 3, 4: the twisted solve of DESIGN 9.1 is the case k = 2) take the
single-problem latency below one CPU core's 0.126 ms?  A cycle model of ONE interior-point iteration of one problem alone on its CU,
built from the per-piece costs the profile builds measured (profiles/r04_wave_phases.txt, r04_twist_latency.txt, r05_wave_phases.txt:
cycles, one wavefront, N = 20) -- no new kernel: the model answers whether one is worth building.

What an exact k-way scheme consists of (Wright 1991; Nielsen & Axehill 2014/15 -- partitioned dynamic programming):
  * the horizon is cut into k partitions of N / k stages; every partition eliminates its own stages at the same time on a wavefront of its
    own.  The FIRST partition knows its entering state (x_0 pinned) and the LAST its leaving cost (none): they are the two halves of the
    twisted solve -- an arrival recursion (1.9 k cycles per stage, measured) and the plain backward recursion (1.55 k per stage).
    An INTERIOR partition knows neither: its elimination carries, next to the 13 x 13 cost block, the 13 x 13 sensitivity to the unknown
    boundary state (the right-hand side of its recursion is a matrix, not a vector): the tile products of a stage (X = P M, G = M'X + C,
    the Schur complement) are done twice -- 2 x the MFMA and gather work of a stage, the 4 x 4 pivot chain once;
  * the k - 1 cuts couple through a block-tridiagonal system of 13 x 13 blocks (the "master problem"): k = 2 is ONE dense 13 x 13 system --
    the meeting system of the twisted solve, measured: factor 6.0 k + solve 2.4 k cycles in the predictor, one more solve (2.4 k, on both
    waves) in the corrector; k - 1 cuts are k - 1 such factorisations IN SEQUENCE (each needs the Schur complement of the one before), or
    a cyclic reduction of depth ceil(log2(k - 1)) + 1 with a 13 x 13 x 13 product per level on top;
  * every partition is back-substituted outwards from its cut states (0.65 k per stage measured for the first-half form, 0.46 k for the
    plain forward sweep);
  * barriers: two more per sweep phase than the plain solve for every level of the master problem (0.6 k each with waves that idle).

    python tools/study/kway_partition_model.py > profiles/r06_kway_model.txt"""
N = 20
# measured, cycles, one problem alone on a CU (profile build)
FACTOR_STAGE = 1550      # plain backward stage (r05: 31.0 k / 20)
ARRIVE_STAGE = 1900      # arrival (first-half) stage (r04: 15.1 k / 8)
PIVOT_CHAIN = 400        # 4 x 4 pivot + hand-over of its factors to the lanes: on the chain once per stage, whatever rides along (r02 ablation)
MEET_FACTOR, MEET_SOLVE = 6000, 2400
FWD_STAGE, BACKSUB_STAGE, BACKVEC_STAGE, ARRVEC_STAGE = 345, 650, 435, 480
BARRIER = 600
EVAL, AFFINE, STEP, NORMS = 7000, 3200, 2600, 1800   # element-wise phases and the termination test (r05, four-wave split)
CLOCK_GHZ = 2.4
ITERATIONS = 4           # BASELINE configs[0] (the drop-in call's problem)
CALL_OVERHEAD_US = 9.5   # launch + completion word (drop-in 125.5 us - kernel 116 us)


def plain():
    pred = N * FACTOR_STAGE + N * FWD_STAGE
    corr = N * BACKVEC_STAGE + N * FWD_STAGE
    return pred, corr


def kway(k, cyclic=False):
    """Predictor / corrector chain of an exact k-way split; the sizes (first, interior ..., last) are the integer split with the shortest
    elimination chain (the first partition's stage costs more than the last's, an interior one's more than either)."""
    INTERIOR_STAGE = 2 * (FACTOR_STAGE - PIVOT_CHAIN) + PIVOT_CHAIN   # tiles twice, pivot chain once
    best = None
    for nf in range(1, N):
        for ni in ([0] if k == 2 else range(1, N)):
            nl = N - nf - (k - 2) * ni
            if nl < 1:
                continue
            e = max(nf * ARRIVE_STAGE, nl * FACTOR_STAGE, ni * INTERIOR_STAGE)
            if best is None or e < best[0]:
                best = (e, nf, ni, nl)
    elim, nf, ni, nl = best
    per = max(nf, ni, nl)
    first, last, interior = nf * ARRIVE_STAGE, nl * FACTOR_STAGE, ni * INTERIOR_STAGE
    cuts = k - 1
    depth = cuts if not cyclic else (1 if cuts == 1 else (2 if cuts <= 3 else 3))
    master_pred = depth * MEET_FACTOR + depth * MEET_SOLVE + (2 * depth) * BARRIER
    master_corr = depth * MEET_SOLVE + (2 * depth) * BARRIER
    backsub = per * max(BACKSUB_STAGE, FWD_STAGE)
    pred = elim + master_pred + backsub
    # corrector: vector sweeps of every partition (the interior ones carry nothing extra: the sensitivity blocks are kept from the predictor)
    corr = per * max(ARRVEC_STAGE, BACKVEC_STAGE) + master_corr + backsub
    return pred, corr, dict(per=per, sizes=(nf, ni, nl), first=first, last=last, interior=interior, master_pred=master_pred, master_corr=master_corr, backsub=backsub)


def iteration(pred, corr):
    return EVAL + NORMS + pred + BARRIER + AFFINE + BARRIER + corr + BARRIER + STEP + BARRIER


def us(cycles):
    return cycles / (CLOCK_GHZ * 1e3)


if __name__ == "__main__":
    p, c = plain()
    it0 = iteration(p, c)
    print(f"{'plain solve':38s}: predictor {p / 1e3:5.1f} k  corrector {c / 1e3:5.1f} k  iteration {it0 / 1e3:5.1f} k cycles (measured 69.5 k)  "
          f"-> {ITERATIONS} iterations {us(ITERATIONS * it0):6.1f} us kernel, call {us(ITERATIONS * it0) + CALL_OVERHEAD_US:6.1f} us (measured 125.5)")
    meas = {2: 101.0}
    for k, cyc in ((2, False), (3, False), (4, False), (4, True), (5, True)):
        p, c, d = kway(k, cyc)
        it = iteration(p, c)
        tag = f"k = {k}{' (cyclic reduction of the cuts)' if cyc else ''}"
        print(f"{tag:38s}: stages per partition (first, interior, last) {d['sizes']}: first {d['first'] / 1e3:4.1f} k | interior {d['interior'] / 1e3:4.1f} k | last {d['last'] / 1e3:4.1f} k; "
              f"cuts {d['master_pred'] / 1e3:4.1f} k (predictor) {d['master_corr'] / 1e3:4.1f} k (corrector); back-substitution {d['backsub'] / 1e3:3.1f} k\n"
              f"{'':38s}  predictor {p / 1e3:5.1f} k  corrector {c / 1e3:5.1f} k  iteration {it / 1e3:5.1f} k ({it / it0:4.2f} x plain)  -> call "
              f"{us(ITERATIONS * it) + CALL_OVERHEAD_US:6.1f} us" + (f"   (built and measured: kernel {meas[k]:.1f} us, call 112-113 us)" if k in meas else ""))
    print("\nRead-out: the cut systems are dense 13 x 13 factorisations of ~4 plain stages each and come IN SEQUENCE; every cut beyond the first costs more than\n"
          "the stages it takes off the longest partition (k = 2 -> 4 with the cuts in sequence: the longest elimination shrinks by ~6 k cycles, the cuts grow by 2 x (8.4 + 1.2) = 19 k; reduced cyclically by 10 k),\n"
          "and an interior partition pays its tile products twice.  k = 2 -- the twisted solve that ships as frp_nmpc_options.twist -- is the optimum of the family on\n"
          "this hardware (0.112-0.113 ms per call measured).  The model is within 2 % of the measured plain kernel and 8 % pessimistic for the twisted one (0.95 x against the\n"
          "measured 0.87 x: its two waves overlap more of the meeting system than the sum above); granting every k the same 8 %, k = 4 with a cyclic reduction of the\n"
          "cuts ties with k = 2 and k = 3, 5 lose -- no k comes near the 0.74 x plain that a\n"
          "0.095 ms call needs (an iteration of 51 k cycles) -- that takes a per-stage factorisation of < 1.1 k cycles, which no partitioning provides.\n"
          "Registers close the same door from the other side: a partition's wavefront holds ~100 VGPRs of tiles and table pointers through its sweep, and the three\n"
          "helper roles of a workgroup carry 66-116 registers of persistent state each at a 168-register cap (DESIGN 4): a third and fourth sweeping wave would spill their\n"
          "state around every sweep -- what the corridor rows on the Riccati wave measured as +8 k cycles per iteration in round 5 (r05_role_splits.txt).")
