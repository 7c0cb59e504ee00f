"""Pmetrics CSV ingest against the reference's own parser tests and data files
(src/data/parser/pmetrics/{mod.rs:438-533, tests.rs, row.rs:717-1172}, src/tests/data/*.csv copied as data
fixtures under tests/golden/pmetrics_*.csv; GOLDEN = tests.rs:9-21)."""
import os

import numpy as np
import pytest

from pharmsol_amd import Bolus, Censor, Data, Infusion, Observation, Subject, interpolate
from pharmsol_amd import _abi, pmetrics
from pharmsol_amd.pmetrics import DataRow, build_data, from_pmetrics_csv_bytes, read_pmetrics

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CORE = ",".join(pmetrics.CORE_HEADERS)


def pm_input(covs, rows):  # tests.rs:23-27
    return (",".join(pmetrics.CORE_HEADERS + list(covs)) + "\n" + rows).encode()


def times(occ):
    return [e.time for e in occ.events]


# --------------------------------------------------------------------------- rows -> events (row.rs:717-1172)
def test_observation_bolus_and_infusion_rows():
    (o,) = DataRow.builder("pt1", 1.0).evid(0).out(25.5).outeq(1).build().into_events()
    assert isinstance(o, Observation) and (o.time, o.value, o.outeq) == (1.0, 25.5, "1")  # label kept as written
    (b,) = DataRow.builder("pt1", 0.0).evid(1).dose(100.0).input(1).build().into_events()
    assert isinstance(b, Bolus) and (b.time, b.amount, b.input) == (0.0, 100.0, "1")
    (i,) = DataRow.builder("pt1", 0.0).evid(1).dose(100.0).dur(2.0).input(1).build().into_events()
    assert isinstance(i, Infusion) and (i.amount, i.duration, i.input) == (100.0, 2.0, "1")


def test_positive_and_negative_addl():
    ev = DataRow.builder("pt1", 0.0).evid(1).dose(100.0).input(1).addl(3).ii(12.0).build().into_events()
    assert [e.time for e in ev] == [12.0, 24.0, 36.0, 0.0]  # additional doses first, then the row's own
    ev = DataRow.builder("pt1", 0.0).evid(1).dose(100.0).input("iv").addl(-3).ii(12.0).build().into_events()
    assert [e.time for e in ev] == [-12.0, -24.0, -36.0, 0.0]
    ev = DataRow.builder("pt1", 0.0).evid(1).dose(100.0).dur(1.5).input("iv").addl(2).ii(24.0).build().into_events()
    assert all(isinstance(e, Infusion) and e.duration == 1.5 for e in ev) and [e.time for e in ev] == [24.0, 48.0, 0.0]


def test_row_errors():
    with pytest.raises(pmetrics.MissingObservationOuteq):
        DataRow.builder("p", 0.0).evid(0).out(1.0).build().into_events()
    with pytest.raises(pmetrics.MissingBolusInput):
        DataRow.builder("p", 0.0).evid(1).dose(1.0).build().into_events()
    with pytest.raises(pmetrics.MissingBolusDose):
        DataRow.builder("p", 0.0).evid(1).input("iv").build().into_events()
    with pytest.raises(pmetrics.MissingInfusionDose):
        DataRow.builder("p", 0.0).evid(1).dur(1.0).input("iv").build().into_events()
    with pytest.raises(pmetrics.UnknownEvid) as e:
        DataRow.builder("p", 0.0).evid(3).build().into_events()
    assert e.value.fields["evid"] == 3
    with pytest.raises(pmetrics.InvalidDataRow, match="negative duration"):
        DataRow.builder("p", 0.0).evid(1).dose(1.0).dur(-1.0).input("iv").build().into_events()
    with pytest.raises(pmetrics.InvalidDataRow, match="requires a dose row"):
        DataRow.builder("p", 0.0).evid(0).out(1.0).outeq("cp").addl(2).ii(1.0).build().into_events()


def test_build_data_doc_example():  # row.rs:573-592
    rows = [DataRow.builder("pt1", 0.0).evid(1).dose(100.0).input("iv").build(),
            DataRow.builder("pt1", 1.0).evid(0).out(50.0).outeq("cp").build(),
            DataRow.builder("pt1", 24.0).evid(4).dose(100.0).input("iv").build(),
            DataRow.builder("pt1", 25.0).evid(0).out(48.0).outeq("cp").build(),
            DataRow.builder("pt2", 0.0).evid(1).dose(50.0).input("iv").build()]
    data = build_data(rows)
    assert len(data.subjects) == 2 and len(data.subjects[0].occasions) == 2
    assert all(e.occasion == 1 for e in data.subjects[0].occasions[1].events)


# --------------------------------------------------------------------------- files (mod.rs:438-533)
def test_addl_file():  # mod.rs:444-490
    data = read_pmetrics(os.path.join(GOLD, "pmetrics_addl_test.csv"))
    s1, s2 = data.subjects
    assert times(s1.occasions[0]) == [-120.0, -108.0, -96.0, -84.0, -72.0, -60.0, -48.0, -36.0, -24.0, -12.0, 0.0, 9.0]
    assert times(s2.occasions[0]) == [0.0, 9.0, 12.0, 24.0, 36.0, 48.0, 60.0, 72.0, 84.0, 96.0, 108.0, 120.0]


def test_named_and_numeric_labels_are_preserved():  # mod.rs:492-532
    d = from_pmetrics_csv_bytes((CORE + "\npt1,1,0,1,100,.,.,iv,.,.,.,.,.,.,.\npt1,0,1,.,.,.,.,.,42,cp,0,.,.,.,.\n").encode())
    ev = d.subjects[0].occasions[0].events
    assert isinstance(ev[0], Infusion) and ev[0].input == "iv" and isinstance(ev[1], Observation) and ev[1].outeq == "cp"
    d = from_pmetrics_csv_bytes((CORE + "\npt1,1,0,.,100,.,.,1,.,.,.,.,.,.,.\npt1,0,1,.,.,.,.,.,42,1,0,.,.,.,.\n").encode())
    ev = d.subjects[0].occasions[0].events
    assert isinstance(ev[0], Bolus) and ev[0].input == "1" and ev[1].outeq == "1"


def test_covariate_file_interpolation():  # covariate.rs:686-760
    data = read_pmetrics(os.path.join(GOLD, "pmetrics_covariate_test.csv"))
    cov = data.subjects[0].occasions[0].covariates
    kn = cov.knots["wt"]
    assert interpolate(kn, 0.0) == 70.0 and interpolate(kn, 24.0) == 72.0 and interpolate(kn, 48.0) == 74.0
    assert abs(interpolate(kn, 12.0) - 70.4) < 1e-8  # knots at 9 h (70) and 24 h (72)
    assert interpolate(kn, 36.0) == 73.0 and interpolate(kn, 60.0) == 74.0


def test_golden_csv_equals_the_builder_fixture():  # tests.rs:9-66 (GOLDEN <-> fixture_data)
    data = read_pmetrics(os.path.join(GOLD, "pmetrics_golden.csv"))
    assert [s.id for s in data.subjects] == ["10", "alpha"]
    ten, alpha = data.subjects
    o0, o1 = ten.occasions
    assert [(type(e).__name__, e.time) for e in o0.events] == [("Observation", 0.0), ("Bolus", 0.0), ("Observation", 1.0),
                                                                ("Infusion", 2.0)]
    first = o0.events[0]
    assert (first.value, first.outeq, first.censoring, first.errorpoly) == (1.25, "cp", Censor.BLOQ, (0.1, 0.2, 0.3, 0.4))
    assert o0.events[1].amount == 100.0 and o0.events[1].input == "iv"
    assert o0.events[2].value is None and o0.events[2].outeq == "2"  # OUT = -99
    assert (o0.events[3].amount, o0.events[3].duration, o0.events[3].input) == (50.0, 4.0, "1")
    assert o0.covariates.knots == {"age": [(0.0, 40.0), (2.0, 41.0)], "wt": [(0.0, 70.0), (2.0, 72.0)]}
    assert o0.covariates.fixed == {"age": True, "wt": False}
    assert [(type(e).__name__, e.time, e.occasion) for e in o1.events] == [("Bolus", 0.0, 1), ("Observation", 3.0, 1)]
    assert (o1.events[1].value, o1.events[1].censoring, o1.events[1].errorpoly) == (9.0, Censor.ALOQ, None)
    assert o1.covariates.knots == {"age": [(0.0, 42.0)], "crcl": [(3.0, 80.0)]} and o1.covariates.fixed["age"] is True
    (a0,) = alpha.occasions
    assert [(type(e).__name__, e.time) for e in a0.events] == [("Observation", 0.0), ("Infusion", 0.0), ("Observation", 2.0)]
    assert a0.events[2].value == -98.5 and a0.events[2].outeq == "neg" and a0.covariates.knots == {"wt": [(0.0, 60.0)]}
    # the same population through the builder (tests.rs:33-66)
    built = (Subject.builder("10").bolus(0.0, 100.0, "iv")
             .observation_with_error(0.0, 1.25, "cp", (0.1, 0.2, 0.3, 0.4), Censor.BLOQ)
             .missing_observation(1.0, "2").infusion(2.0, 50.0, "1", 4.0).build())
    assert [(type(e).__name__, e.time) for e in built.occasions[0].events] == [(type(e).__name__, e.time) for e in o0.events]


# --------------------------------------------------------------------------- header / field rules (tests.rs:193-283)
def test_existing_files_and_placeholders_remain_readable():
    d = from_pmetrics_csv_bytes(pm_input(["WT"], "s,1,0,0,1,.,.,iv,.,.,.,.,.,.,.,70\n"))
    assert "wt" in d.subjects[0].occasions[0].covariates.knots
    from_pmetrics_csv_bytes(pm_input([], "s,0,0,0,0,0,0,unused,1,cp,0,0,0,0,0\ns,1,1,0,1,0,0,iv,-99,unused,0,0,0,0,0\n"))
    from_pmetrics_csv_bytes(pm_input([], "s,0,0,.,.,.,.,.,.,cp,0,.,.,.,.\ns,0,1,.,.,.,.,.,NA,cp,0,.,.,.,.\ns,0,2,.,.,.,.,.,,cp,0,.,.,.,.\n"))


@pytest.mark.parametrize("tail", [",WT,wt\n", ",WT!,wt!\n", ",WT,wt!\n", ",wt!!\n", ",wt!x\n"])
def test_duplicate_and_conflicting_headers_are_rejected(tail):
    with pytest.raises(pmetrics.InvalidPmetricsData):
        from_pmetrics_csv_bytes((CORE + tail).encode())


def test_duplicate_core_header_is_rejected():
    with pytest.raises(pmetrics.InvalidPmetricsData, match="duplicate core header"):
        from_pmetrics_csv_bytes(("id," + CORE + "\n").encode())


@pytest.mark.parametrize("text,missing", [("", "ID"), ("EVID,TIME\n", "ID"), ("ID,TIME\n", "EVID"), ("ID,EVID\n", "TIME")])
def test_required_core_headers_are_validated_without_data_rows(text, missing):
    with pytest.raises(pmetrics.InvalidPmetricsData, match=f"missing required core header `{missing}`"):
        from_pmetrics_csv_bytes(text.encode())


def test_unused_core_headers_may_be_omitted():
    d = from_pmetrics_csv_bytes(b"ID,EVID,TIME,DOSE,INPUT\ns,1,0,100,iv\n")
    assert isinstance(d.subjects[0].occasions[0].events[0], Bolus)
    d = from_pmetrics_csv_bytes(b"ID,EVID,TIME,OUT,OUTEQ\ns,0,0,1.5,cp\n")
    assert isinstance(d.subjects[0].occasions[0].events[0], Observation)


def test_mixed_case_covariate_headers_are_normalized():
    d = from_pmetrics_csv_bytes(pm_input(["WT!", "Ka"], "s,1,0,0,1,.,.,iv,.,.,.,.,.,.,.,70,0.5\n"))
    cov = d.subjects[0].occasions[0].covariates
    assert cov.fixed["wt"] is True and cov.fixed["ka"] is False and cov.knots["ka"] == [(0.0, 0.5)]


# --------------------------------------------------------------------------- ADDL + EVID=4 (tests.rs:444-563)
def test_negative_addl_reset_starts_the_occasion_at_the_earliest_expanded_dose():
    d = from_pmetrics_csv_bytes(pm_input([], "s,1,0,0,1,.,.,iv,.,.,.,.,.,.,.\ns,4,0,0,2,-2,1,iv,.,.,.,.,.,.,.\n"))
    occ = d.subjects[0].occasions[1]
    assert times(occ) == [-2.0, -1.0, 0.0] and all(e.occasion == 1 for e in occ.events)
    d = from_pmetrics_csv_bytes(pm_input([], "s,1,0,0,1,.,.,iv,.,.,.,.,.,.,.\ns,4,2,0,2,-2,1,iv,.,.,.,.,.,.,.\n"))
    assert times(d.subjects[0].occasions[1]) == [0.0, 1.0, 2.0]


@pytest.mark.parametrize("ii", [".", "0", "-1"])
def test_nonzero_addl_requires_positive_ii(ii):
    with pytest.raises(pmetrics.InvalidDataRow, match="requires a positive II"):
        from_pmetrics_csv_bytes((CORE + f"\ns,1,0,0,1,2,{ii},iv,.,.,.,.,.,.,.\n").encode())


def test_minimum_addl_and_time_overflow_fail_cleanly():
    with pytest.raises(pmetrics.InvalidDataRow, match="too large to expand"):
        from_pmetrics_csv_bytes((CORE + f"\ns,1,0,0,1,{-2 ** 63},1,iv,.,.,.,.,.,.,.\n").encode())
    with pytest.raises(pmetrics.NonFiniteValue) as e:
        from_pmetrics_csv_bytes((CORE + "\ns,1,0,0,1,2,1e308,iv,.,.,.,.,.,.,.\n").encode())
    assert e.value.fields["field"] == "expanded TIME"


# --------------------------------------------------------------------------- covariates in rows (tests.rs:567-620)
def test_covariate_values_at_one_time():
    d = from_pmetrics_csv_bytes(pm_input(["wt"], "s,1,0,0,1,.,.,iv,.,.,.,.,.,.,.,70\ns,0,0,.,.,.,.,.,1,cp,0,.,.,.,.,70\n"))
    assert d.subjects[0].occasions[0].covariates.knots["wt"] == [(0.0, 70.0)]
    with pytest.raises(pmetrics.InvalidDataRow) as e:
        from_pmetrics_csv_bytes(pm_input(["wt"], "s,1,0,0,1,.,.,iv,.,.,.,.,.,.,.,70\ns,0,0,.,.,.,.,.,1,cp,0,.,.,.,.,71\n"))
    msg = str(e.value)
    assert "conflicting covariate `wt` values" in msg and "subject `s` occasion 0" in msg and "time 0" in msg
    d = from_pmetrics_csv_bytes(pm_input(["wt"], "s,1,0,0,1,.,.,iv,.,.,.,.,.,.,.,70\ns,0,24,.,.,.,.,.,1,cp,0,.,.,.,.,72\n"))
    kn = d.subjects[0].occasions[0].covariates.knots["wt"]
    assert kn == [(0.0, 70.0), (24.0, 72.0)] and interpolate(kn, 12.0) == 71.0


# --------------------------------------------------------------------------- rejected inputs (tests.rs:657-676, 796-874)
def test_evid_2_empty_reset_and_empty_id():
    with pytest.raises(pmetrics.UnknownEvid) as e:
        from_pmetrics_csv_bytes(pm_input(["wt"], "s,2,0,.,.,.,.,.,.,.,.,.,.,.,.,70\n"))
    assert e.value.fields["evid"] == 2 and e.value.fields["id"] == "s"
    with pytest.raises(pmetrics.InvalidDataRow, match="must contain a dose"):
        from_pmetrics_csv_bytes((CORE + "\ns,4,0,.,.,.,.,.,.,.,.,.,.,.,.\n").encode())
    with pytest.raises(pmetrics.InvalidDataRow, match="subject ID cannot be empty"):
        from_pmetrics_csv_bytes((CORE + "\n,0,0,.,.,.,.,.,1,cp,0,.,.,.,.\n").encode())


def test_malformed_and_nonfinite_values_fail_cleanly():
    with pytest.raises(pmetrics.InvalidDataRow):  # partial error polynomial
        from_pmetrics_csv_bytes((CORE + "\ns,0,0,.,.,.,.,.,1,cp,0,0.1,.,.,.\n").encode())
    with pytest.raises(pmetrics.NonFiniteValue):
        from_pmetrics_csv_bytes((CORE + "\ns,1,NaN,0,1,.,.,iv,.,.,.,.,.,.,.\n").encode())
    with pytest.raises(pmetrics.CSVError):
        from_pmetrics_csv_bytes((CORE + "\ns,1,zero,0,1,.,.,iv,.,.,.,.,.,.,.\n").encode())
    with pytest.raises(pmetrics.CSVError):
        from_pmetrics_csv_bytes((CORE + "\ns,1,0,0,1,.,.,iv,.,.,maybe,.,.,.,.\n").encode())


def test_comment_lines_and_subject_order():
    text = "# exported\n" + CORE + "\nzed,1,0,0,1,.,.,iv,.,.,.,.,.,.,.\n# mid-file note\nabe,1,0,0,2,.,.,iv,.,.,.,.,.,.,.\n"
    d = Data.from_pmetrics_csv_bytes(text.encode())
    assert [s.id for s in d.subjects] == ["abe", "zed"]  # sorted by ID (row.rs:671)


# --------------------------------------------------------------------------- the parsed population feeds the path
def test_parsed_population_flattens_through_a_model():
    from pharmsol_amd import Ratio, analytical, bolus, infusion

    eq = analytical(name="one_cmt", params=["ke", "v"], structure="one_compartment", states=["central"], outputs=["cp"],
                    routes=[bolus("1", "central"), infusion("1", "central")], out={"cp": Ratio("central", "v")})
    data = from_pmetrics_csv_bytes(
        b"ID,EVID,TIME,DUR,DOSE,ADDL,II,INPUT,OUT,OUTEQ\n"
        b"a,1,0,0,600,2,12,1,.,.\na,0,9,.,.,.,.,.,10,cp\na,0,30,.,.,.,.,.,-99,cp\n"
        b"b,1,0,1.5,300,.,.,1,.,.\nb,0,2,.,.,.,.,.,4,cp\nb,4,24,0,300,.,.,1,.,.\nb,0,26,.,.,.,.,.,3,cp\n")
    flat = eq.flatten(data)
    assert flat.n_subjects == 2 and flat.n_observations == 4 and flat.n_events == 9
    assert flat.n_occasions == 3 and list(flat.occ_index) == [0, 0, 1]
    vals = flat.ev_value[flat.ev_kind == _abi.PMX_EV_OBSERVATION]
    assert np.isnan(vals[1]) and list(vals[[0, 2, 3]]) == [10.0, 4.0, 3.0]
