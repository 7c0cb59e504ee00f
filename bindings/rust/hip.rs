//! Safe layer over `pmx_sys.rs` — the file a pharmsol maintainer drops in as `src/simulator/hip.rs`.
//!
//! NOT COMPILED IN THIS REPOSITORY'S BUILD IMAGE (no rustc / cargo there).  What is checked instead
//! (tests/test_rust_binding.py): every `pmx_*` call below names a function of `pmx_sys.rs` with the
//! declared number of arguments, every `PMX_*` constant exists, and the `pmx_population_desc { .. }`
//! literal names every field of the struct once, in the header's order.  The same ABI is exercised end
//! to end by the ctypes binding (`pharmsol_amd/_ffi.py`).
//!
//! What it replaces, in the reference's terms:
//!
//! * `flatten(eq, data)`          – `Data` → the SoA descriptor, labels resolved ONCE with
//!                                  `Equation::resolve_input_label / resolve_output_label`
//!                                  (src/simulator/equation/mod.rs:195-245), instead of once per
//!                                  (subject, support point) in `resolve_occasion_events` (mod.rs:247-273).
//! * `HipData`                    – the device-resident dataset, built once per fit.
//! * `HipAnalytical` / `HipOde`   – newtypes whose population entries are ONE library call each:
//!                                  `log_likelihood_matrix` (src/simulator/likelihood/matrix.rs:52-106),
//!                                  `log_likelihood_batch` (likelihood/mod.rs:119-177), and the prediction
//!                                  matrix of the loop nest matrix.rs:79-98.
//! * `PinnedMatrix`               – page-locked S x P output the DMA engine writes at link rate.
//! * `HipShards`                  – one population per GPU (contiguous subject ranges balanced by events),
//!                                  the in-place all-gather of the prediction blocks over RCCL.
#![allow(non_camel_case_types)]

use std::ffi::{c_void, CStr};
use std::marker::PhantomData;
use std::ptr;

use ndarray::{Array2, ArrayView2};

use crate::data::{Censor, Data, Event, Subject};
use crate::simulator::equation::{Equation, RouteKind};
use crate::PharmsolError;

#[path = "pmx_sys.rs"]
mod pmx_sys;
use pmx_sys::*;

// ------------------------------------------------------------------------------------------ errors
fn check(rc: i32) -> Result<(), PharmsolError> {
    if rc == PMX_OK {
        return Ok(());
    }
    // thread-local text set by the failing call (include/pmx.h "errors")
    let msg = unsafe { CStr::from_ptr(pmx_last_error()) }.to_string_lossy().into_owned();
    Err(match rc {
        // Equation::validate_event_bounds (equation/mod.rs:322-327 and :333-338)
        PMX_ERR_INPUT_OUT_OF_RANGE | PMX_ERR_OUTEQ_OUT_OF_RANGE | PMX_ERR_PAIR_FAILED => PharmsolError::OtherError(msg),
        _ => PharmsolError::OtherError(format!("libpmx_hip ({rc}): {msg}")),
    })
}

/// Called once per process before the first handle is made: the binding and the library must agree on the ABI
/// version and on every struct's size (`pmx_sizeof_struct`, include/pmx.h "versioning").
pub fn verify_abi() -> Result<(), PharmsolError> {
    let ok = unsafe {
        pmx_abi_version() == PMX_ABI_VERSION
            && pmx_sizeof_struct(c"pmx_population_desc".as_ptr()) as usize == std::mem::size_of::<pmx_population_desc>()
            && pmx_sizeof_struct(c"pmx_model_desc".as_ptr()) as usize == std::mem::size_of::<pmx_model_desc>()
            && pmx_sizeof_struct(c"pmx_error_model".as_ptr()) as usize == std::mem::size_of::<pmx_error_model>()
    };
    if ok {
        Ok(())
    } else {
        Err(PharmsolError::OtherError("libpmx_hip: ABI mismatch between pmx_sys.rs and the loaded library".into()))
    }
}

// ------------------------------------------------------------------------------------------ flatten
/// `Data` as the structure-of-arrays `pmx_population_desc` points into.  Owns the arrays; `desc()` borrows them.
pub struct FlatData {
    subj_occ_off: Vec<i64>,
    occ_ev_off: Vec<i64>,
    occ_index: Vec<i32>,
    ev_time: Vec<f64>,
    ev_value: Vec<f64>,
    ev_duration: Vec<f64>,
    ev_kind: Vec<u8>,
    ev_io: Vec<u16>,
    n_covariates: i32,
    cov_knot_off: Vec<i64>,
    cov_knot_time: Vec<f64>,
    cov_knot_value: Vec<f64>,
    cov_fixed: Vec<u8>,
    ev_errorpoly: Vec<f64>,
    ev_censor: Vec<i8>,
}

/// One pass over `data.subjects() -> occasions() -> events()`.  `covariates`: the model's covariate names in the
/// order its closures index them (the macro's declaration order; `ValidatedModelMetadata` has them for macro and
/// DSL models, a hand-written model passes the order its closures read).
pub fn flatten<E: Equation>(eq: &E, data: &Data, covariates: &[&str]) -> Result<FlatData, PharmsolError> {
    let mut f = FlatData {
        subj_occ_off: vec![0],
        occ_ev_off: vec![0],
        occ_index: Vec::new(),
        ev_time: Vec::new(),
        ev_value: Vec::new(),
        ev_duration: Vec::new(),
        ev_kind: Vec::new(),
        ev_io: Vec::new(),
        n_covariates: covariates.len() as i32,
        cov_knot_off: vec![0],
        cov_knot_time: Vec::new(),
        cov_knot_value: Vec::new(),
        cov_fixed: Vec::new(),
        ev_errorpoly: Vec::new(),
        ev_censor: Vec::new(),
    };
    for subject in data.iter() {
        flatten_subject(eq, subject, covariates, &mut f)?;
        f.subj_occ_off.push(f.occ_index.len() as i64);
    }
    Ok(f)
}

fn flatten_subject<E: Equation>(eq: &E, subject: &Subject, covariates: &[&str], f: &mut FlatData) -> Result<(), PharmsolError> {
    for occasion in subject.occasions() {
        f.occ_index.push(occasion.index() as i32); // init runs for index 0 only (analytical/mod.rs:417)
        for event in occasion.events() {
            // labels -> dense indices, the kind-aware rule of equation/mod.rs:195-233
            let (kind, io, value, duration, poly, censor) = match event {
                Event::Bolus(b) => (PMX_EV_BOLUS, eq.resolve_input_label(b.input(), RouteKind::Bolus)?, b.amount(), 0.0, None, Censor::None),
                Event::Infusion(i) => {
                    (PMX_EV_INFUSION, eq.resolve_input_label(i.input(), RouteKind::Infusion)?, i.amount(), i.duration(), None, Censor::None)
                }
                Event::Observation(o) => (
                    PMX_EV_OBSERVATION,
                    eq.resolve_output_label(o.outeq())?,
                    o.value().unwrap_or(f64::NAN), // None = prediction only (event.rs:575-582)
                    0.0,
                    o.errorpoly(),
                    o.censoring(),
                ),
            };
            f.ev_time.push(event.time());
            f.ev_value.push(value);
            f.ev_duration.push(duration);
            f.ev_kind.push(kind as u8);
            f.ev_io.push(io as u16);
            match poly {
                Some(p) => f.ev_errorpoly.extend_from_slice(&[p.c0(), p.c1(), p.c2(), p.c3()]),
                None => f.ev_errorpoly.extend_from_slice(&[f64::NAN, 0.0, 0.0, 0.0]), // c0 = NaN: the error model's polynomial
            }
            f.ev_censor.push(match censor {
                Censor::None => PMX_CENSOR_NONE,
                Censor::BLOQ => PMX_CENSOR_BLOQ,
                Censor::ALOQ => PMX_CENSOR_ALOQ,
            } as i8);
        }
        f.occ_ev_off.push(f.ev_time.len() as i64);
        // raw knots per (occasion, covariate): the library interpolates like Covariate::interpolate (covariate.rs:189-241)
        for name in covariates {
            match occasion.covariates().get_covariate(name) {
                Some(cov) => {
                    for (t, v) in cov.observations() {
                        f.cov_knot_time.push(t);
                        f.cov_knot_value.push(v);
                    }
                    f.cov_fixed.push(cov.fixed() as u8);
                }
                None => f.cov_fixed.push(0), // no knots: the device sees what `MissingSegments` sees - NaN, failed pair
            }
            f.cov_knot_off.push(f.cov_knot_time.len() as i64);
        }
    }
    Ok(())
}

impl FlatData {
    pub fn n_subjects(&self) -> usize {
        self.subj_occ_off.len() - 1
    }
    /// The descriptor, borrowing `self` (the library copies what it needs inside `pmx_population_create`).
    pub fn desc(&self) -> pmx_population_desc {
        let cov = self.n_covariates > 0;
        pmx_population_desc {
            n_subjects: self.n_subjects() as i64,
            n_occasions: self.occ_index.len() as i64,
            n_events: self.ev_time.len() as i64,
            subj_occ_off: self.subj_occ_off.as_ptr(),
            occ_ev_off: self.occ_ev_off.as_ptr(),
            occ_index: self.occ_index.as_ptr(),
            ev_time: self.ev_time.as_ptr(),
            ev_value: self.ev_value.as_ptr(),
            ev_duration: self.ev_duration.as_ptr(),
            ev_kind: self.ev_kind.as_ptr(),
            ev_io: self.ev_io.as_ptr(),
            n_covariates: self.n_covariates,
            presorted: 1, // Subject::new sorted every occasion already (structs.rs:363-369)
            cov_knot_off: if cov { self.cov_knot_off.as_ptr() } else { ptr::null() },
            cov_knot_time: if cov { self.cov_knot_time.as_ptr() } else { ptr::null() },
            cov_knot_value: if cov { self.cov_knot_value.as_ptr() } else { ptr::null() },
            cov_fixed: if cov { self.cov_fixed.as_ptr() } else { ptr::null() },
            ev_errorpoly: self.ev_errorpoly.as_ptr(),
            ev_censor: self.ev_censor.as_ptr(),
        }
    }
}

// ------------------------------------------------------------------------------------------ device dataset
/// Device-resident dataset: built once, reused by every cycle of the fit.
pub struct HipData {
    pop: *mut pmx_population,
    n_obs: usize,
    n_subjects: usize,
}
unsafe impl Send for HipData {}
unsafe impl Sync for HipData {} // the library serialises the per-handle workspaces itself (DESIGN.md "threads")

impl HipData {
    pub fn new<E: Equation>(eq: &E, data: &Data, covariates: &[&str], device: i32) -> Result<Self, PharmsolError> {
        verify_abi()?;
        let flat = flatten(eq, data, covariates)?;
        Self::from_flat(&flat, None, device)
    }
    /// `subjects = Some((s0, s1))`: only that contiguous subject range is compiled and uploaded (one rank's shard).
    pub fn from_flat(flat: &FlatData, subjects: Option<(i64, i64)>, device: i32) -> Result<Self, PharmsolError> {
        let desc = flat.desc();
        let mut pop = ptr::null_mut();
        match subjects {
            None => check(unsafe { pmx_population_create(&desc, device, &mut pop) })?,
            Some((s0, s1)) => check(unsafe { pmx_population_create_shard(&desc, s0, s1, device, &mut pop) })?,
        }
        Ok(Self {
            pop,
            n_obs: unsafe { pmx_population_n_observations(pop) } as usize,
            n_subjects: unsafe { pmx_population_n_subjects(pop) } as usize,
        })
    }
    pub fn n_observations(&self) -> usize {
        self.n_obs
    }
    pub fn n_subjects(&self) -> usize {
        self.n_subjects
    }
    /// Row r of a prediction matrix belongs to `subject[r]`, at `time[r]`, output `outeq[r]` - the order of
    /// `SubjectPredictions::flat_predictions()` (src/simulator/likelihood/subject.rs:145-148), subject after subject.
    pub fn observation_info(&self) -> Result<(Vec<f64>, Vec<i32>, Vec<i64>), PharmsolError> {
        let (mut t, mut o, mut s) = (vec![0.0; self.n_obs], vec![0i32; self.n_obs], vec![0i64; self.n_obs]);
        check(unsafe { pmx_population_observation_info(self.pop, t.as_mut_ptr(), o.as_mut_ptr(), s.as_mut_ptr()) })?;
        Ok((t, o, s))
    }
}
impl Drop for HipData {
    fn drop(&mut self) {
        unsafe { pmx_population_destroy(self.pop) }
    }
}

// ------------------------------------------------------------------------------------------ pinned output
/// Row-major `[rows x cols]` doubles in page-locked host memory (`pmx_host_alloc`): the host-pointer entry points
/// recognise it and let the DMA engine write it directly.
pub struct PinnedMatrix {
    p: *mut f64,
    rows: usize,
    cols: usize,
}
unsafe impl Send for PinnedMatrix {}

impl PinnedMatrix {
    pub fn zeros(rows: usize, cols: usize) -> Result<Self, PharmsolError> {
        let mut raw: *mut c_void = ptr::null_mut();
        check(unsafe { pmx_host_alloc((rows * cols * 8).max(8) as i64, &mut raw) })?;
        unsafe { ptr::write_bytes(raw as *mut u8, 0, rows * cols * 8) };
        Ok(Self { p: raw as *mut f64, rows, cols })
    }
    pub fn view(&self) -> ArrayView2<'_, f64> {
        unsafe { ArrayView2::from_shape_ptr((self.rows, self.cols), self.p) }
    }
    pub fn as_mut_ptr(&mut self) -> *mut f64 {
        self.p
    }
    pub fn shape(&self) -> (usize, usize) {
        (self.rows, self.cols)
    }
}
impl Drop for PinnedMatrix {
    fn drop(&mut self) {
        unsafe { pmx_host_free(self.p as *mut c_void) }
    }
}

// ------------------------------------------------------------------------------------------ models
/// What the two back-ends share: a model handle and the population entry points.
struct HipModel {
    model: *mut pmx_model,
}
unsafe impl Send for HipModel {}
unsafe impl Sync for HipModel {} // handles are immutable after creation (Equation: Sync, equation/mod.rs:377)

impl HipModel {
    fn from_desc(desc: &pmx_model_desc) -> Result<Self, PharmsolError> {
        verify_abi()?;
        let mut model = ptr::null_mut();
        check(unsafe { pmx_model_create(desc, &mut model) })?;
        Ok(Self { model })
    }
    /// Closures as C/HIP source text (`pmx_derive`, `pmx_route_lag`, `pmx_route_bioavailability`, `pmx_init`, `pmx_outputs`,
    /// `pmx_seq_eq`, `pmx_eq`, `pmx_dynamics`, `pmx_dynamics_bolus`; `functions` = the PMX_FN_* bits the text defines).
    fn from_source(desc: &pmx_model_desc, source: &CStr, functions: u32) -> Result<Self, PharmsolError> {
        verify_abi()?;
        let mut model = ptr::null_mut();
        check(unsafe { pmx_model_create_user(desc, source.as_ptr(), functions, &mut model) })?;
        Ok(Self { model })
    }
    fn theta_rows<'a>(theta: &'a Array2<f64>) -> ndarray::CowArray<'a, f64, ndarray::Ix2> {
        theta.as_standard_layout() // row-major [P x k] (matrix.rs:62-65)
    }
    fn log_likelihood_matrix(&self, data: &HipData, theta: &Array2<f64>, em: &[pmx_error_model], psi: &mut PinnedMatrix)
        -> Result<(), PharmsolError> {
        let th = Self::theta_rows(theta);
        let p = th.nrows();
        assert_eq!(psi.shape(), (data.n_subjects(), p));
        check(unsafe { pmx_loglik(self.model, data.pop, em.as_ptr(), th.as_ptr(), p as i64, psi.as_mut_ptr(), p as i64, ptr::null_mut()) })
    }
    fn log_likelihood_batch(&self, data: &HipData, parameters: &Array2<f64>, em: &[pmx_error_model]) -> Result<Vec<f64>, PharmsolError> {
        let th = Self::theta_rows(parameters);
        assert_eq!(th.nrows(), data.n_subjects()); // one parameter row per subject (likelihood/mod.rs:126-133)
        let mut ll = vec![0.0; data.n_subjects()];
        check(unsafe { pmx_loglik_batch(self.model, data.pop, em.as_ptr(), th.as_ptr(), ll.as_mut_ptr(), ptr::null_mut()) })?;
        Ok(ll) // a failed subject scores -inf, like the reference's batch (likelihood/mod.rs:165-170)
    }
    fn predictions_matrix(&self, data: &HipData, theta: &Array2<f64>) -> Result<(Array2<f64>, Vec<u8>), PharmsolError> {
        let th = Self::theta_rows(theta);
        let p = th.nrows();
        let mut pred = Array2::<f64>::zeros((data.n_observations(), p));
        let mut status = vec![0u8; data.n_subjects() * p];
        let rc = unsafe { pmx_predict(self.model, data.pop, th.as_ptr(), p as i64, pred.as_mut_ptr(), p as i64, status.as_mut_ptr()) };
        if rc != PMX_ERR_PAIR_FAILED {
            check(rc)?; // PAIR_FAILED: NaN rows + status bytes tell which (subject, support point) failed
        }
        Ok((pred, status))
    }
}
impl Drop for HipModel {
    fn drop(&mut self) {
        unsafe { pmx_model_destroy(self.model) }
    }
}

macro_rules! hip_backend {
    ($name:ident, $eq_kind:expr, $doc:literal) => {
        #[doc = $doc]
        pub struct $name {
            inner: HipModel,
            _not_clone: PhantomData<*mut ()>,
        }
        unsafe impl Send for $name {}
        unsafe impl Sync for $name {}

        impl $name {
            /// From the descriptor the `analytical!` / `ode!` lowering fills.
            pub fn from_desc(mut desc: pmx_model_desc) -> Result<Self, PharmsolError> {
                desc.eq_kind = $eq_kind;
                Ok(Self { inner: HipModel::from_desc(&desc)?, _not_clone: PhantomData })
            }
            /// From closure source text, compiled for gfx950 by hiprtc inside the library.
            pub fn from_source(mut desc: pmx_model_desc, source: &CStr, functions: u32) -> Result<Self, PharmsolError> {
                desc.eq_kind = $eq_kind;
                Ok(Self { inner: HipModel::from_source(&desc, source, functions)?, _not_clone: PhantomData })
            }
            /// `log_likelihood_matrix(&eq, &data, &theta, &error_models, _)` (likelihood/matrix.rs:52-106): THE call of
            /// an NPAG cycle.  Predictions never leave the GPU; S x P doubles come back into `psi`.
            pub fn log_likelihood_matrix(&self, data: &HipData, theta: &Array2<f64>, em: &[pmx_error_model], psi: &mut PinnedMatrix)
                -> Result<(), PharmsolError> {
                self.inner.log_likelihood_matrix(data, theta, em, psi)
            }
            /// `log_likelihood_batch(&eq, &data, &parameters, &residual_error_models)` (likelihood/mod.rs:119-177),
            /// `em[o].kind = PMX_EM_RES_*`.
            pub fn log_likelihood_batch(&self, data: &HipData, parameters: &Array2<f64>, em: &[pmx_error_model])
                -> Result<Vec<f64>, PharmsolError> {
                self.inner.log_likelihood_batch(data, parameters, em)
            }
            /// `estimate_predictions` for every subject x every support point (the loop nest of matrix.rs:79-98):
            /// `pred[row, p]`, rows in `HipData::observation_info` order, plus the per-(subject, support point) status.
            pub fn estimate_predictions_matrix(&self, data: &HipData, theta: &Array2<f64>) -> Result<(Array2<f64>, Vec<u8>), PharmsolError> {
                self.inner.predictions_matrix(data, theta)
            }
        }
    };
}
hip_backend!(HipAnalytical, PMX_EQ_ANALYTICAL, "`Analytical` (src/simulator/equation/analytical/mod.rs) whose population entries run on the GPU.");
hip_backend!(HipOde, PMX_EQ_ODE, "`ODE` (src/simulator/equation/ode/mod.rs) whose population entries run on the GPU.");

/// `AssayErrorModel::{Additive, Proportional}` (src/data/error_model.rs:786-812) as the library's record.
pub fn assay_error_model(em: &crate::data::error_model::AssayErrorModel) -> Result<pmx_error_model, PharmsolError> {
    use crate::data::error_model::AssayErrorModel as A;
    let (kind, poly, scalar) = match em {
        A::Additive { .. } => (PMX_EM_ADDITIVE, em.errorpoly()?, em.factor()?),
        A::Proportional { .. } => (PMX_EM_PROPORTIONAL, em.errorpoly()?, em.factor()?),
        A::None => return Ok(pmx_error_model { kind: PMX_EM_NONE, reserved: 0, c: [0.0; 4], scalar: 0.0 }),
    };
    Ok(pmx_error_model { kind, reserved: 0, c: [poly.c0(), poly.c1(), poly.c2(), poly.c3()], scalar })
}

// ------------------------------------------------------------------------------------------ more than one GPU
/// One process per GPU.  Every rank flattens the same `Data` (or rank 0 broadcasts the flat arrays), asks the library
/// for the split, and keeps only its own subject range on its device.  Subjects are independent, so prediction and
/// log-likelihood passes need no exchange; only a caller that wants the WHOLE prediction matrix on every rank gathers.
pub struct HipShards {
    pub bounds: Vec<i64>, // [n_ranks + 1] subject ranges, balanced by event count (pmx_shard_bounds)
    pub rows: Vec<i64>,   // [n_ranks + 1] first prediction row of each rank's block
    pub local: HipData,
    comm: *mut pmx_comm,
}
unsafe impl Send for HipShards {}

impl HipShards {
    /// `unique_id`: PMX_COMM_ID_BYTES made by rank 0 with `HipShards::unique_id()` and handed to the other ranks by
    /// whatever launched them (MPI, a file, the environment); `None` = no collective wanted.
    pub fn new(flat: &FlatData, n_ranks: i32, rank: i32, device: i32, unique_id: Option<&[u8]>) -> Result<Self, PharmsolError> {
        let desc = flat.desc();
        let mut bounds = vec![0i64; n_ranks as usize + 1];
        let mut rows = vec![0i64; n_ranks as usize + 1];
        check(unsafe { pmx_shard_bounds(&desc, n_ranks, bounds.as_mut_ptr()) })?;
        check(unsafe { pmx_shard_rows(&desc, n_ranks, bounds.as_ptr(), rows.as_mut_ptr()) })?;
        let local = HipData::from_flat(flat, Some((bounds[rank as usize], bounds[rank as usize + 1])), device)?;
        let mut comm = ptr::null_mut();
        if let Some(id) = unique_id {
            assert_eq!(id.len(), PMX_COMM_ID_BYTES as usize);
            check(unsafe { pmx_comm_create(id.as_ptr(), n_ranks, rank, device, &mut comm) })?;
        }
        Ok(Self { bounds, rows, local, comm })
    }
    pub fn unique_id() -> Result<Vec<u8>, PharmsolError> {
        let mut id = vec![0u8; PMX_COMM_ID_BYTES as usize];
        check(unsafe { pmx_comm_unique_id(id.as_mut_ptr()) })?;
        Ok(id)
    }
    /// `d_full`: DEVICE buffer of `rows[n_ranks] x ld` doubles; this rank has written its block at row `rows[rank]`
    /// (`pmx_predict_device` with `d_pred = d_full + rows[rank] * ld`).  In place, on `stream`.
    ///
    /// # Safety
    /// `d_full` must be a device allocation of that size on this rank's device.
    pub unsafe fn allgather_predictions(&self, d_full: *mut f64, ld: i64, stream: *mut c_void) -> Result<(), PharmsolError> {
        check(pmx_allgather_predictions(self.comm, d_full, self.rows.as_ptr(), ld, stream))
    }
}
impl Drop for HipShards {
    fn drop(&mut self) {
        if !self.comm.is_null() {
            unsafe { pmx_comm_destroy(self.comm) }
        }
    }
}

// build.rs of the crate that includes this file:
//   println!("cargo:rustc-link-search=native={}/pharmsol_amd/lib", repo_root);
//   println!("cargo:rustc-link-lib=dylib=pmx_hip");
