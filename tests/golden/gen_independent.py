#!/usr/bin/env python3
"""Generate tests/golden/independent_math.json: independent-mathematics fixtures for the
compartment models (SURVEY.md §8c "independent-math fixtures").

Each case = (structure, theta, schedule) -> predictions computed WITHOUT the oracle and without
the reference: the linear ODE system dx/dt = A x + b(t) is propagated exactly between all
breakpoints (event times and infusion starts/ends) with the augmented matrix exponential
expm([[A, b], [0, 0]] * dt) in mpmath at 40 digits, boluses added at their times, and
y = x[central]/v read at observation times.  Event semantics used here are the documented ones
(observation before dose at equal times; infusion rate amount/duration on [t, t+dur)).

Run:  python tests/golden/gen_independent.py     (deterministic; numpy Generator seed 20261003)
"""
import json
import os

import mpmath as mp
import numpy as np

mp.mp.dps = 40
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "independent_math.json")

# structure -> (n_states, builder(theta) -> (A, central, gut_or_None), n_rate_params)
def rate_matrix(structure, th):
    if structure == "one_compartment":
        ke, = th[:1]
        return [[-ke]], 0
    if structure == "one_compartment_with_absorption":
        ka, ke = th[:2]
        return [[-ka, 0], [ka, -ke]], 1
    if structure == "two_compartments":
        ke, kcp, kpc = th[:3]
        return [[-(ke + kcp), kpc], [kcp, -kpc]], 0
    if structure == "two_compartments_with_absorption":
        ke, ka, kcp, kpc = th[:4]
        return [[-ka, 0, 0], [ka, -(ke + kcp), kpc], [0, kcp, -kpc]], 1
    if structure == "three_compartments":
        k10, k12, k13, k21, k31 = th[:5]
        return [[-(k10 + k12 + k13), k21, k31], [k12, -k21, 0], [k13, 0, -k31]], 0
    if structure == "three_compartments_with_absorption":
        ka, k10, k12, k13, k21, k31 = th[:6]
        return [[-ka, 0, 0, 0], [ka, -(k10 + k12 + k13), k21, k31], [0, k12, -k21, 0], [0, k13, 0, -k31]], 1
    raise KeyError(structure)


NPAR = {"one_compartment": 1, "one_compartment_with_absorption": 2, "two_compartments": 3,
        "two_compartments_with_absorption": 4, "three_compartments": 5, "three_compartments_with_absorption": 6}


def simulate(structure, theta, events):
    """events: list of (kind, time, value, duration, io) with kind in {"obs","bolus","inf"}; returns predictions
    in the documented event order (time, then obs < bolus < inf, stable)."""
    A, central = rate_matrix(structure, [mp.mpf(float(t)) for t in theta])
    n = len(A)
    v = mp.mpf(float(theta[-1]))
    rank = {"obs": 0, "bolus": 1, "inf": 2}
    ev = sorted(events, key=lambda e: (e[1], rank[e[0]]))
    infs = [(mp.mpf(e[1]), mp.mpf(e[1]) + mp.mpf(e[3]), mp.mpf(e[2]) / mp.mpf(e[3])) for e in ev if e[0] == "inf"]
    x = mp.matrix(n, 1)
    t = mp.mpf(ev[0][1]) if ev else mp.mpf(0)

    def advance(x, t0, t1):
        pts = sorted({t0, t1} | {b for (s, e, _) in infs for b in (s, e) if t0 < b < t1})
        for a, b in zip(pts[:-1], pts[1:]):
            rate = sum((r for (s, e, r) in infs if s <= a and b <= e), mp.mpf(0))
            M = mp.matrix(n + 1, n + 1)
            for i in range(n):
                for j in range(n):
                    M[i, j] = A[i][j]
            M[central, n] = rate  # infusions enter the central compartment (rateiv[0])
            E = mp.expm(M * (b - a))
            xa = mp.matrix(n + 1, 1)
            for i in range(n):
                xa[i] = x[i]
            xa[n] = 1
            xb = E * xa
            x = mp.matrix([xb[i] for i in range(n)])
        return x

    preds = []
    for e in ev:
        te = mp.mpf(e[1])
        if te > t:
            x = advance(x, t, te)
            t = te
        if e[0] == "bolus":
            x[int(e[4])] += mp.mpf(e[2])
        elif e[0] == "obs":
            preds.append(float(x[central] / v))
    return preds


def random_theta(rng, structure):
    k = NPAR[structure]
    th = list(np.exp(rng.uniform(np.log(0.05), np.log(2.0), size=k)))
    if "absorption" in structure:
        # keep ka well away from the elimination eigenvalues (the closed forms are singular at ka == lambda_i)
        ka_idx = 1 if structure == "two_compartments_with_absorption" else 0
        th[ka_idx] = float(rng.uniform(4.0, 8.0))
    return th + [float(rng.uniform(5.0, 80.0))]


def random_events(rng, structure):
    n_states = len(rate_matrix(structure, [1.0] * 8)[0])
    has_gut = "absorption" in structure
    ev = []
    for _ in range(int(rng.integers(1, 4))):
        t = float(np.round(rng.uniform(0, 30), 2))
        if rng.random() < 0.5:
            io = 0 if (has_gut and rng.random() < 0.7) else (1 if has_gut else 0)
            ev.append(("bolus", t, float(np.round(rng.uniform(20, 400), 1)), 0.0, io))
        else:
            ev.append(("inf", t, float(np.round(rng.uniform(20, 400), 1)), float(np.round(rng.uniform(0.25, 5), 2)), 0))
    for _ in range(int(rng.integers(3, 9))):
        ev.append(("obs", float(np.round(rng.uniform(0, 48), 2)), 0.0, 0.0, 0))
    return ev


def main():
    rng = np.random.default_rng(20261003)
    cases = []
    for structure in NPAR:
        for _ in range(12):
            th = random_theta(rng, structure)
            ev = random_events(rng, structure)
            cases.append({"structure": structure, "theta": th, "events": [list(e) for e in ev],
                          "expected": simulate(structure, th, ev)})
    with open(OUT, "w") as f:
        json.dump({"generator": "tests/golden/gen_independent.py", "mp_dps": 40, "cases": cases}, f, indent=1)
    print(f"wrote {len(cases)} cases to {OUT}")


if __name__ == "__main__":
    main()
