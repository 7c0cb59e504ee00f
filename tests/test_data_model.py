"""Host data model: builder, sort order, repeat/reset, label resolution, Parameters — mirrors the
reference's unit tests in src/data/builder.rs:364-515, src/data/event.rs:774-884, src/parameters.rs."""
import numpy as np
import pytest

from pharmsol_amd import (Bolus, Data, Infusion, LabelError, Observation, Parameters, Ratio, Subject, analytical, bolus,
                          infusion, _abi)
from pharmsol_amd.data import interpolate
from tests import models


def test_subject_builder_basic():  # builder.rs:369-391
    s = (Subject.builder("s1").observation(3.0, 100.0, 0).repeat(2, 0.5).bolus(1.0, 100.0, 0)
         .infusion(0.0, 100.0, 0, 1.0).repeat(3, 1.0).covariate("c1", 0.0, 5.0).covariate("c1", 5.0, 10.0).reset()
         .observation(10.0, 100.0, 0).bolus(7.0, 100.0, 0).repeat(4, 1.0).build())
    assert len(s.occasions) == 2
    assert len(s.occasions[0].events) == 3 + 1 + 4
    assert len(s.occasions[1].events) == 1 + 5
    assert s.occasions[0].covariates.knots["c1"] == [(0.0, 5.0), (5.0, 10.0)]
    assert s.occasions[1].index == 1


def test_events_sorted_by_time_then_type():  # event.rs:292-304
    s = (Subject.builder("x").infusion(1.0, 1.0, 0, 1.0).bolus(1.0, 1.0, 0).observation(1.0, 0.0, 0)
         .observation(0.5, 0.0, 0).bolus(0.5, 2.0, 0).build())
    kinds = [type(e).__name__ for e in s.occasions[0].events]
    assert kinds == ["Observation", "Bolus", "Observation", "Bolus", "Infusion"]


def test_equal_time_and_type_keep_insertion_order():
    s = Subject.builder("x").bolus(1.0, 1.0, 0).bolus(1.0, 2.0, 0).bolus(1.0, 3.0, 0).build()
    assert [e.amount for e in s.occasions[0].events] == [1.0, 2.0, 3.0]


def test_repeat_spacing():  # builder.rs:424-455
    s = Subject.builder("x").infusion(0.0, 100.0, 0, 1.0).repeat(3, 12.0).build()
    assert [e.time for e in s.occasions[0].events] == [0.0, 12.0, 24.0, 36.0]
    assert all(isinstance(e, Infusion) and e.duration == 1.0 for e in s.occasions[0].events)


def test_missing_observation_has_no_value():
    s = Subject.builder("x").missing_observation(1.0, "cp").observation(2.0, 3.5, "cp").build()
    assert s.occasions[0].events[0].value is None and s.occasions[0].events[1].value == 3.5


def _named_model():
    return analytical(name="m", params=["ka", "ke", "v"], structure="one_compartment_with_absorption",
                      states=["gut", "central"], outputs=["cp"],
                      routes=[bolus("oral", "gut"), infusion("iv", "central")], out={"cp": Ratio("central", "v")})


def test_routes_are_numbered_per_kind():  # metadata.rs:926-946; analytical/mod.rs:742-750
    m = _named_model()
    assert m.resolve_input_label("oral", "bolus") == 0
    assert m.resolve_input_label("iv", "infusion") == 0
    assert m.ndrugs == 1


def test_label_resolution_errors():  # equation/mod.rs:195-245
    m = _named_model()
    with pytest.raises(LabelError, match="UnsupportedInputRouteKind"):
        m.resolve_input_label("iv", "bolus")
    with pytest.raises(LabelError, match="unknown input label"):
        m.resolve_input_label("nope", "bolus")
    with pytest.raises(LabelError, match="unknown output label"):
        m.resolve_output_label("conc")
    with pytest.raises(LabelError):
        m.resolve_input_label("0", "bolus")  # a bare number never falls back to a declaration position


def test_numeric_aliases_resolve_against_canonical_labels():  # analytical/mod.rs:603-653
    m = analytical(name="alias", params=["ke", "v"], structure="one_compartment", states=["central"],
                   outputs=["outeq_1"], routes=[infusion("input_1", "central")], out={"outeq_1": Ratio("central", "v")})
    assert m.resolve_input_label("input_1", "infusion") == m.resolve_input_label("1", "infusion") == 0
    assert m.resolve_output_label("outeq_1") == m.resolve_output_label("1") == 0


def test_without_metadata_labels_must_be_dense_indices():
    m = models.handwritten_analytical("one_compartment", 0, 2)
    assert m.resolve_input_label(1, "bolus") == 1 and m.resolve_output_label("0") == 0
    with pytest.raises(LabelError):
        m.resolve_input_label("oral", "bolus")


def test_parameters_with_model():  # parameters.rs:74-91
    m = _named_model()
    p = Parameters.with_model(m, [("v", 50.0), ("ka", 1.2), ("ke", 0.1)])
    np.testing.assert_array_equal(p.as_slice(), [1.2, 0.1, 50.0])
    with pytest.raises(KeyError):
        Parameters.with_model(m, [("ka", 1.0), ("ke", 0.1)])
    with pytest.raises(KeyError):
        Parameters.with_model(m, [("ka", 1.0), ("ke", 0.1), ("v", 1.0), ("zz", 2.0)])


def test_macro_identity_binding_vs_projection():  # expand/analytical.rs:213
    ident = _named_model().desc()
    assert ident.n_bind == 0
    m = analytical(name="p", params=["v", "ke", "ka"], structure="one_compartment_with_absorption",
                   states=["gut", "central"], outputs=["cp"], routes=[bolus("oral", "gut")],
                   out={"cp": Ratio("central", "v")})
    d = m.desc()
    assert d.n_bind == 2 and (d.bind[0].index, d.bind[1].index) == (2, 1)


def test_flatten_layout():
    m = _named_model()
    s1 = Subject.builder("a").bolus(0.0, 100.0, "oral").missing_observation(1.0, "cp").build()
    s2 = (Subject.builder("b").infusion(0.0, 50.0, "iv", 2.0).missing_observation(1.0, "cp").reset()
          .missing_observation(0.5, "cp").build())
    f = m.flatten(Data([s1, s2]))
    assert (f.n_subjects, f.n_occasions, f.n_events, f.n_observations) == (2, 3, 5, 3)
    np.testing.assert_array_equal(f.subj_occ_off, [0, 1, 3])
    np.testing.assert_array_equal(f.occ_ev_off, [0, 2, 4, 5])
    np.testing.assert_array_equal(f.occ_index, [0, 0, 1])
    np.testing.assert_array_equal(f.ev_kind, [_abi.PMX_EV_BOLUS, _abi.PMX_EV_OBSERVATION, _abi.PMX_EV_INFUSION,
                                              _abi.PMX_EV_OBSERVATION, _abi.PMX_EV_OBSERVATION])
    np.testing.assert_array_equal(f.observation_offsets(), [0, 1, 3])
    sl = f.subject_slice(1, 2)
    assert (sl.n_subjects, sl.n_occasions, sl.n_events) == (1, 2, 3)
    np.testing.assert_array_equal(sl.occ_ev_off, [0, 2, 3])


def test_missing_covariate_is_an_error():  # fetch_cov! panics (src/lib.rs:433-443)
    m = models.readme_analytical()
    s = Subject.builder("nocov").bolus(0.0, 1.0, "oral").missing_observation(1.0, "cp").build()
    with pytest.raises(KeyError, match="Covariate wt not found"):
        m.flatten(s)


def test_python_interpolate_matches_reference_rules():  # covariate.rs:216-241
    kn = [(0.0, 70.0), (10.0, 80.0), (20.0, 60.0)]
    assert interpolate(kn, -1.0) == 70.0 and interpolate(kn, 25.0) == 60.0 and interpolate(kn, 20.0) == 60.0
    assert abs(interpolate(kn, 5.0) - 75.0) < 1e-12 and interpolate(kn, 5.0, fixed=True) == 70.0
