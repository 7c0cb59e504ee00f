"""Model and subject fixtures shared by the tests (re-creations of the reference's own test fixtures)."""
from __future__ import annotations

import numpy as np

from pharmsol_amd import (ODE, Analytical, Parameters, Pow, Ratio, Scaled, Subject, analytical, bolus, infusion, ode)


def readme_analytical():
    """examples/analytical_readme.rs:7-24"""
    return analytical(name="one_cmt_oral", params=["ka", "ke0", "v"],
                      derived={"ke": Scaled("ke0", (Pow("wt", 70.0, 0.75),))}, covariates=["wt"],
                      states=["gut", "central"], outputs=["cp"], routes=[bolus("oral", "gut")],
                      structure="one_compartment_with_absorption", out={"cp": Ratio("central", "v")})


def readme_subject():
    """examples/analytical_readme.rs:26-33"""
    return (Subject.builder("analytical_readme").bolus(0.0, 500.0, "oral").missing_observation(0.5, "cp")
            .missing_observation(1.0, "cp").missing_observation(2.0, "cp").missing_observation(4.0, "cp")
            .covariate("wt", 0.0, 75.0).build())


def infusion_dosing_subject():
    """SubjectInfo::InfusionDosing, analytical/mod.rs:446-462"""
    b = Subject.builder("id1").bolus(0.0, 100.0, 0).infusion(24.0, 150.0, 0, 3.0)
    for t in (0.0, 1.0, 2.0, 4.0, 8.0, 12.0, 24.0, 25.0, 26.0, 27.0, 28.0, 32.0, 36.0):
        b = b.missing_observation(t, 0)
    return b.build()


def oral_infusion_subject():
    """SubjectInfo::OralInfusionDosage, analytical/mod.rs:464-487"""
    b = Subject.builder("id1").bolus(0.0, 100.0, 1).infusion(24.0, 150.0, 0, 3.0).bolus(48.0, 100.0, 0)
    for t in (0.0, 1.0, 2.0, 4.0, 8.0, 12.0, 24.0, 25.0, 26.0, 27.0, 28.0, 32.0, 36.0, 48.0, 49.0, 50.0, 52.0, 56.0,
              60.0):
        b = b.missing_observation(t, 0)
    return b.build()


# (structure, central state, theta incl. trailing v, subject factory, matching built-in diffeq or None)
# parameter sets from the reference's kernel unit tests:
#   one_compartment_models.rs:96,151 / two_compartment_models.rs:165,225 / three_compartment_models.rs:304,371
KERNEL_CASES = [
    ("one_compartment", 0, [0.1, 1.0], infusion_dosing_subject, "one_cmt_iv"),
    ("one_compartment_with_absorption", 1, [1.0, 0.1, 1.0], oral_infusion_subject, "one_cmt_oral"),
    ("two_compartments", 0, [0.1, 3.0, 1.0, 1.0], infusion_dosing_subject, "two_cmt_iv"),
    ("two_compartments_with_absorption", 1, [0.1, 1.0, 3.0, 1.0, 1.0], oral_infusion_subject, "two_cmt_oral"),
    ("three_compartments", 0, [0.1, 3.0, 2.0, 1.0, 0.5, 1.0], infusion_dosing_subject, "three_cmt_iv"),
    ("three_compartments_with_absorption", 1, [1.0, 0.1, 3.0, 2.0, 1.0, 0.5, 1.0], oral_infusion_subject,
     "three_cmt_oral"),
    # CL forms (no ODE twin): *_cl_models.rs tests use the same subjects
    ("one_compartment_cl", 0, [0.1, 1.0, 1.0], infusion_dosing_subject, None),
    ("one_compartment_cl_with_absorption", 1, [1.0, 0.1, 1.0, 1.0], oral_infusion_subject, None),
    ("two_compartments_cl", 0, [0.1, 3.0, 1.0, 2.0, 1.0], infusion_dosing_subject, None),
    ("two_compartments_cl_with_absorption", 1, [1.0, 0.1, 3.0, 1.0, 2.0, 1.0], oral_infusion_subject, None),
    ("three_compartments_cl", 0, [0.1, 3.0, 2.0, 1.0, 3.0, 4.0, 1.0], infusion_dosing_subject, None),
    ("three_compartments_cl_with_absorption", 1, [1.0, 0.1, 3.0, 2.0, 1.0, 3.0, 4.0, 1.0], oral_infusion_subject,
     None),
]


def handwritten_analytical(structure: str, central: int, nparams: int):
    """`Analytical::new(kernel, ..).with_nstates(n).with_ndrugs(2).with_nout(1)` with out = x[central]/theta[last]."""
    from pharmsol_amd import _abi

    ns = _abi.KERNEL_STATE_COUNT[structure]
    return Analytical.new(structure, {0: Ratio(central, nparams - 1)}, nparams=nparams).with_nstates(ns).with_ndrugs(
        2).with_nout(1)


def handwritten_ode(diffeq: str, central: int, nparams: int, h_max: float = 0.01):
    from pharmsol_amd import _abi

    ns = _abi.ODE_STATE_COUNT[diffeq]
    return ODE.new(diffeq, {0: Ratio(central, nparams - 1)}, nparams=nparams, h_max=h_max).with_nstates(ns).with_ndrugs(
        2).with_nout(1)


def random_subject(rng: np.random.Generator, n_bolus_inputs: int = 1, multi_occasion: bool = False,
                   ties: bool = True) -> Subject:
    """A ragged subject: random boluses/infusions (possibly overlapping) and observations, with deliberate
    time ties (observation at a dose time, dose at an infusion end)."""
    b = Subject.builder(f"r{rng.integers(1 << 30)}")
    n_occ = 1 + (int(rng.integers(0, 3)) if multi_occasion else 0)
    for occ in range(n_occ):
        if occ > 0:
            b = b.reset()
        n_dose = int(rng.integers(1, 5))
        marks = []
        for _ in range(n_dose):
            t = float(np.round(rng.uniform(0, 48), 1)) if ties else float(rng.uniform(0, 48))
            if rng.random() < 0.5:
                b = b.bolus(t, float(rng.uniform(50, 500)), int(rng.integers(0, n_bolus_inputs)))
                marks.append(t)
            else:
                d = float(np.round(rng.uniform(0.1, 6), 1)) if ties else float(rng.uniform(0.1, 6))
                b = b.infusion(t, float(rng.uniform(50, 500)), 0, d)
                marks += [t, t + d]
        n_obs = int(rng.integers(1, 12))
        for _ in range(n_obs):
            if ties and marks and rng.random() < 0.3:
                t = marks[int(rng.integers(0, len(marks)))]
            else:
                t = float(rng.uniform(0, 72))
            b = b.missing_observation(t, 0)
    return b.build()
