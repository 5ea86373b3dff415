#!/usr/bin/env python3
"""How do the 64 lanes of k_roots_e share a wave?  A CPU model.

The oracle's eigenvalue iteration (oracle/linalg.h built with -DORACLE_EIG_TRACE, the sink in the usage line below) records,
for every degree-10 polynomial of real 5-point samples, its sequence of events -- one eigenvalue deflated, a 2 x 2 block
split off, a Francis step over the bulge positions imm .. iu-2.  This script replays 64 such sequences in lockstep the way
the register-resident kernel executes them (per iteration of the wave: the window search, then the union of the branch
bodies its lanes need, the Francis body over the union of their windows) with instruction weights read off the kernel's ISA,
and reports the lane utilisation of the present schedule and of two variants.

    g++ -O2 -fPIC -shared -DORACLE_EIG_TRACE -Ioracle -o /tmp/liboracle_trace.so oracle/*.cc tools/eig_trace_sink.cc
    python tools/sim_roots_lanes.py /tmp/liboracle_trace.so"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dagsfm_amd import synthetic  # noqa: E402
from tests import oracle_lib  # noqa: E402

C_SEARCH, C_D1, C_D2, C_FINIT, C_POS, C_FTAIL = [int(x) for x in os.environ.get("SIM_WEIGHTS", "60,40,450,300,230,100").split(",")]


def traces(lib_path, n_pairs=30, trials=64, seed=0):
    L = ctypes.CDLL(lib_path)
    L.oracle_estimate_model.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    L.oracle_eig_trace_take.argtypes = [ctypes.c_void_p, ctypes.c_int]
    orc = oracle_lib.load()
    scene = synthetic.Scene(10, 2048, seed=seed)
    ims = [scene.image(i) for i in range(10)]
    rng = np.random.default_rng(seed)
    out = []
    buf = np.zeros(4096, np.int32)
    models = np.zeros(90)
    k = 0
    for i in range(10):
        for j in range(i + 1, 10):
            if k >= n_pairs:
                break
            k += 1
            m = np.asarray(orc.match_sift_features_cpu(ims[i][0], ims[j][0])).reshape(-1, 2)
            if len(m) < 10:
                continue
            c = np.array([scene.width / 2, scene.height / 2])
            p1 = (ims[i][1][m[:, 0]].astype(np.float64) - c) / scene.focal
            p2 = (ims[j][1][m[:, 1]].astype(np.float64) - c) / scene.focal
            for _ in range(trials):
                s = rng.choice(len(m), 5, replace=False)
                a, b = np.ascontiguousarray(p1[s]), np.ascontiguousarray(p2[s])
                L.oracle_eig_trace_take(buf.ctypes.data, 0)
                L.oracle_estimate_model(4, a.ctypes.data, b.ctypes.data, 5, models.ctypes.data)
                n = L.oracle_eig_trace_take(buf.ctypes.data, len(buf))
                ev = [(int(v) & 15, (int(v) >> 4) & 15, (int(v) >> 8) & 15, (int(v) >> 12) & 15) for v in buf[:n]]
                if ev:
                    out.append(ev)
    return out


def lane_cost(ev):
    c = 0
    for kind, il, imm, iu in ev:
        c += C_SEARCH
        if kind == 1:
            c += C_D1
        elif kind == 2:
            c += C_D2
        else:
            c += C_FINIT + C_POS * (iu - 1 - imm) + C_FTAIL
    return c


def kinds_of(wave, pos):
    return [wave[l][pos[l]][0] if pos[l] < len(wave[l]) else 0 for l in range(len(wave))]


def francis(wave, pos, kinds):
    ks = set()
    for l, k in enumerate(kinds):
        if k == 3:
            _, il, imm, iu = wave[l][pos[l]]
            ks.update(range(imm, iu - 1))
            pos[l] += 1
    return C_FINIT + C_POS * len(ks) + C_FTAIL


def simulate(wave, fuse_deflations):
    """fuse_deflations: inside one iteration of the wave a lane first works off its pending deflations (search + deflation
    bodies, up to three rounds) and then takes its Francis step -- instead of one event per lane and iteration."""
    pos = [0] * len(wave)
    cost = iters = 0
    while any(p < len(e) for p, e in zip(pos, wave)):
        iters += 1
        if fuse_deflations:
            for _ in range(3):
                kinds = kinds_of(wave, pos)
                if not any(k in (1, 2) for k in kinds):
                    break
                cost += C_SEARCH + (C_D1 if 1 in kinds else 0) + (C_D2 if 2 in kinds else 0)
                for l, k in enumerate(kinds):
                    if k in (1, 2):
                        pos[l] += 1
            kinds = kinds_of(wave, pos)
            cost += C_SEARCH
            if 3 in kinds:
                cost += francis(wave, pos, kinds)
        else:
            kinds = kinds_of(wave, pos)
            cost += C_SEARCH + (C_D1 if 1 in kinds else 0) + (C_D2 if 2 in kinds else 0)
            for l, k in enumerate(kinds):
                if k in (1, 2):
                    pos[l] += 1
            if 3 in kinds:
                cost += francis(wave, pos, kinds)
    return cost, iters


def simulate_capped(wave, cap):
    """The present schedule for at most `cap` iterations of the wave; returns the cost and what is left of every lane's sequence."""
    pos = [0] * len(wave)
    cost = iters = 0
    while iters < cap and any(p < len(e) for p, e in zip(pos, wave)):
        iters += 1
        kinds = kinds_of(wave, pos)
        cost += C_SEARCH + (C_D1 if 1 in kinds else 0) + (C_D2 if 2 in kinds else 0)
        for l, k in enumerate(kinds):
            if k in (1, 2):
                pos[l] += 1
        if 3 in kinds:
            cost += francis(wave, pos, kinds)
    return cost, [e[p:] for p, e in zip(pos, wave) if p < len(e)]


def two_launches(tr, cap, rng, n_waves=60):
    """A capped first launch, the unfinished polynomials compacted into full waves for a second one (the round-5 verdict's item 6)."""
    cost1, rest = 0, []
    for w in range(n_waves):
        idx = rng.choice(len(tr), 64, replace=False)
        c, r = simulate_capped([tr[i] for i in idx], cap)
        cost1 += c
        rest += r
    cost2 = 0
    for w in range(0, len(rest), 64):
        cost2 += simulate(rest[w:w + 64], False)[0] * (len(rest[w:w + 64]) / 64.0 if len(rest[w:w + 64]) < 64 else 1.0)
    return (cost1 + cost2) / n_waves, len(rest) / (64.0 * n_waves)


def main():
    tr = traces(sys.argv[1] if len(sys.argv) > 1 else "/tmp/liboracle_trace.so")
    cnt = lambda e, k: sum(1 for x in e if x[0] == k)  # noqa: E731
    print("polynomials traced: %d; events each: %.1f (Francis steps %.1f, single deflations %.1f, 2 x 2 deflations %.1f); bulge positions each %.1f" % (
        len(tr), np.mean([len(e) for e in tr]), np.mean([cnt(e, 3) for e in tr]), np.mean([cnt(e, 1) for e in tr]),
        np.mean([cnt(e, 2) for e in tr]), np.mean([sum(x[3] - 1 - x[2] for x in e if x[0] == 3) for e in tr])))
    fr = np.array([cnt(e, 3) for e in tr])
    print("Francis steps per polynomial: percentiles 10/50/90/99/max = %s" % np.percentile(fr, [10, 50, 90, 99, 100]).tolist())
    ideal = np.mean([lane_cost(e) for e in tr])
    rng = np.random.default_rng(1)
    for name, fuse in (("present schedule (one event per lane and iteration)", False), ("deflations fused into the iteration", True)):
        costs, its = [], []
        for w in range(60):
            idx = rng.choice(len(tr), 64, replace=False)
            c, it = simulate([tr[i] for i in idx], fuse)
            costs.append(c)
            its.append(it)
        print("%-60s wave cost %.0f (one lane's own work %.0f: utilisation %.2f), iterations %.1f" % (name, np.mean(costs), ideal, ideal / np.mean(costs), np.mean(its)))
    base = None
    for cap in (0, 14, 16, 18, 20, 22, 24, 26):
        c, frac = two_launches(tr, cap if cap else 10 ** 6, np.random.default_rng(1))
        if not cap:
            base = c
        print("two launches, the first capped at %2s iterations: wave cost %.0f (utilisation %.2f, %+.1f %% against one launch), %.0f %% of the polynomials go on to the second" % (
            cap if cap else "no", c, ideal / c, 100.0 * (base / c - 1.0), 100.0 * frac))
    order = np.argsort([len(e) for e in tr])
    costs = []
    for w in range(0, len(order) - 63, 64):
        costs.append(simulate([tr[i] for i in order[w:w + 64]], True)[0])
    print("%-60s wave cost %.0f (utilisation %.2f)" % ("fused + waves of polynomials with equal event counts (bound)", np.mean(costs), ideal / np.mean(costs)))


if __name__ == "__main__":
    main()
