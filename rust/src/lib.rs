//! Rust binding of `include/kmc.h` (libkmc.so).  NOT COMPILED in the build environment of this
//! repository (no rustc there); a CPU test parses this file and checks every declaration against the
//! header (tests/test_abi_host.py::test_rust_binding_matches_the_header).
use std::ffi::CStr;
use std::os::raw::{c_char, c_int, c_void};

pub const KMC_MODE_CONTIG: i32 = 0;
pub const KMC_MODE_LR: i32 = 1;
pub const KMC_ALGO_AUTO: i32 = 0;
pub const KMC_ALGO_STREAM: i32 = 1;
pub const KMC_ALGO_WALK: i32 = 2;
pub const KMC_ALGO_SORT: i32 = 3;
pub const KMC_FORGET_MEMO: c_int = 1;
pub const KMC_FORGET_HISTORY: c_int = 2;

// The structs and the extern block below are checked against include/kmc.h by
// tests/test_abi_host.py::test_rust_binding_matches_the_header (names, arity, argument types, field
// order and types), because no Rust toolchain exists in the build environment to do it.
#[repr(C)]
pub struct KmcConfig {
    pub struct_size: u32,
    pub k: i32,
    pub mode: i32,
    pub canonical: i32,
    pub device: i32,
    pub algo: i32,
    pub capacity_hint: u64,
    pub stream: *mut c_void,
}

#[repr(C)]
pub struct KmcStats {
    pub n_reads: u64,
    pub n_bases: u64,
    pub n_kmers: u64,
    pub n_distinct: u64,
    pub table_capacity: u64,
    pub n_spilled: u64,
    pub n_batches: u64,
    pub kernel_ms_last: f64,
    pub kernel_ms_total: f64,
    pub algo_last: i32,
    pub launches_last: i32,
    pub n_slabs_skipped: u64,
    pub n_direct: u64,
    pub kernel_ms_lifetime: f64,
    pub launches_lifetime: u64,
    pub n_async_ok: u64,
    pub n_async_slabs_skipped: u64,
    pub n_planner_stale: u64,
}

#[repr(C)]
pub struct KmcReads {
    pub bases: *mut u8,
    pub offsets: *mut u64,
    pub n_reads: u64,
    pub n_bases: u64,
    pub max_read_len: u64,
}

#[repr(C)]
pub struct KmcSynth {
    pub seed: u64,
    pub pool: u32,
    pub line_len: u32,
    pub lines_per_record: u32,
    pub reserved: u32,
}

#[repr(C)]
pub struct KmcCtx {
    _private: [u8; 0],
}

#[repr(C)]
pub struct KmcFastaStream {
    _private: [u8; 0],
}

extern "C" {
    pub fn kmc_version() -> *const c_char;
    pub fn kmc_status_string(status: c_int) -> *const c_char;
    pub fn kmc_create(out: *mut *mut KmcCtx, cfg: *const KmcConfig) -> c_int;
    pub fn kmc_destroy(ctx: *mut KmcCtx);
    pub fn kmc_last_error(ctx: *const KmcCtx) -> *const c_char;
    pub fn kmc_reset(ctx: *mut KmcCtx) -> c_int;
    pub fn kmc_add_batch(ctx: *mut KmcCtx, bases: *const u8, offsets: *const u64, n_reads: u64) -> c_int;
    pub fn kmc_add_batch_device(ctx: *mut KmcCtx, d_bases: *const c_void, d_offsets: *const c_void, n_reads: u64, n_bases: u64, max_read_len: u64) -> c_int;
    pub fn kmc_merge_pairs_device(ctx: *mut KmcCtx, d_key_hi: *const c_void, d_key_lo: *const c_void, d_count: *const c_void, n_pairs: u64) -> c_int;
    pub fn kmc_finalize(ctx: *mut KmcCtx, n_distinct: *mut u64, n_total: *mut u64) -> c_int;
    pub fn kmc_export(ctx: *mut KmcCtx, key_hi: *mut u64, key_lo: *mut u64, count: *mut u64, cap: u64) -> c_int;
    pub fn kmc_export_device(ctx: *mut KmcCtx, d_key_hi: *mut *const c_void, d_key_lo: *mut *const c_void, d_count: *mut *const c_void, n_distinct: *mut u64) -> c_int;
    pub fn kmc_partition_device(ctx: *mut KmcCtx, n_parts: u32, part_begin: *mut u64, d_key_hi: *mut *const c_void,
                                d_key_lo: *mut *const c_void, d_count: *mut *const c_void) -> c_int;
    pub fn kmc_owner_of(key_hi: u64, key_lo: u64, n_parts: u32) -> u32;
    // multi-GPU reduce (one process per GPU; the collective itself is the host program's, e.g. RCCL)
    pub fn kmc_slab_words(ctx: *const KmcCtx, slab_entries: u64) -> u64;
    pub fn kmc_pack_slab_device(ctx: *mut KmcCtx, d_slab: *mut c_void, slab_entries: u64) -> c_int;
    pub fn kmc_merge_slabs_device(ctx: *mut KmcCtx, d_slabs: *const c_void, n_slabs: u32, slab_entries: u64, my_part: u32, n_parts: u32) -> c_int;
    pub fn kmc_poll(ctx: *mut KmcCtx) -> c_int;
    pub fn kmc_sync(ctx: *mut KmcCtx) -> c_int;
    pub fn kmc_finalize_async(ctx: *mut KmcCtx) -> c_int;
    pub fn kmc_read_peak_device(d_buf: *const c_void, n_bytes: u64, device: c_int, stream: *mut c_void, shape: c_int, iters: c_int, ms_avg: *mut f64, xor_out: *mut u64) -> c_int;
    pub fn kmc_read_pieces(read_len: u64, k: c_int, starts: *mut u64, ends: *mut u64, cap: u64) -> u64;
    pub fn kmc_forget_source(ctx: *mut KmcCtx, what: c_int) -> c_int;
    pub fn kmc_get_stats(ctx: *const KmcCtx, out: *mut KmcStats) -> c_int;
    pub fn kmc_count_file(ctx: *mut KmcCtx, path: *const c_char, n_distinct: *mut u64, n_total: *mut u64) -> c_int;
    pub fn kmc_count_file_multi(ctxs: *mut *mut KmcCtx, n_ctx: u32, path: *const c_char, n_distinct: *mut u64, n_total: *mut u64) -> c_int;
    // host reader: whole file, or streaming (chunks end at record boundaries; buffers owned by the stream)
    pub fn kmc_parse_fasta(path: *const c_char, out: *mut KmcReads, errbuf: *mut c_char, errbuf_len: usize) -> c_int;
    pub fn kmc_free_reads(r: *mut KmcReads);
    pub fn kmc_fasta_stream_open(path: *const c_char, chunk_bytes: u64, out: *mut *mut KmcFastaStream, errbuf: *mut c_char, errbuf_len: usize) -> c_int;
    pub fn kmc_fasta_stream_next(s: *mut KmcFastaStream, out: *mut KmcReads, eof: *mut c_int, errbuf: *mut c_char, errbuf_len: usize) -> c_int;
    pub fn kmc_fasta_stream_close(s: *mut KmcFastaStream);
    pub fn kmc_decode_key(key_hi: u64, key_lo: u64, klen: c_int, out: *mut c_char);
    // seeded re-creation of random_fasta_generator.py's distribution
    pub fn kmc_synth_records_for_bytes(s: *const KmcSynth, file_bytes: u64, exact_bytes: *mut u64) -> u64;
    pub fn kmc_synth_reads_host(s: *const KmcSynth, first_record: u64, n_records: u64, bases: *mut u8, offsets: *mut u64) -> c_int;
    pub fn kmc_synth_reads_device(s: *const KmcSynth, first_record: u64, n_records: u64, d_bases: *mut c_void, d_offsets: *mut c_void, device: c_int, stream: *mut c_void) -> c_int;
    pub fn kmc_synth_write_fasta(s: *const KmcSynth, first_record: u64, n_records: u64, FILE_ptr: *mut c_void) -> c_int;
}

/// One counting context on one GPU.  Not `Sync`: a ctx is single-threaded (kmc.h).
pub struct Counter {
    ctx: *mut KmcCtx,
    klen: i32,
}

#[derive(Debug)]
pub struct KmcError(pub i32, pub String);

impl Counter {
    /// `k = None` is the reference's own computation (27 + gap + 27, sizes 80..=140, main.rs:48-49,63).
    pub fn new(k: Option<i32>, canonical: bool, device: i32) -> Result<Counter, KmcError> {
        let cfg = KmcConfig {
            struct_size: std::mem::size_of::<KmcConfig>() as u32,
            k: k.unwrap_or(54),
            mode: if k.is_some() { KMC_MODE_CONTIG } else { KMC_MODE_LR },
            canonical: canonical as i32,
            device,
            algo: KMC_ALGO_AUTO,
            capacity_hint: 0,
            stream: std::ptr::null_mut(),
        };
        let mut ctx = std::ptr::null_mut();
        let rc = unsafe { kmc_create(&mut ctx, &cfg) };
        if rc != 0 {
            let msg = unsafe { CStr::from_ptr(kmc_last_error(std::ptr::null())) }.to_string_lossy().into_owned();
            return Err(KmcError(rc, msg));
        }
        Ok(Counter { ctx, klen: k.unwrap_or(54) })
    }

    fn check(&self, rc: c_int) -> Result<(), KmcError> {
        if rc == 0 {
            return Ok(());
        }
        let msg = unsafe { CStr::from_ptr(kmc_last_error(self.ctx)) }.to_string_lossy().into_owned();
        Err(KmcError(rc, msg))
    }

    /// `bases`: all reads concatenated; `offsets[n_reads+1]`.  Buffers are free again on return.
    pub fn add_batch(&mut self, bases: &[u8], offsets: &[u64]) -> Result<(), KmcError> {
        if offsets.is_empty() {
            return Err(KmcError(-1, "offsets must hold n_reads + 1 entries (at least one)".into()));
        }
        let rc = unsafe { kmc_add_batch(self.ctx, bases.as_ptr(), offsets.as_ptr(), (offsets.len() - 1) as u64) };
        self.check(rc)
    }

    /// FASTA path in: the library's own pipelined reader (parse on the host cores overlapped with
    /// upload and counting) instead of `bio`'s record loop (main.rs:44-46,58-62).
    pub fn count_file(&mut self, path: &str) -> Result<(u64, u64), KmcError> {
        let c = std::ffi::CString::new(path).map_err(|_| KmcError(-1, "path contains NUL".into()))?;
        let (mut nd, mut nt) = (0u64, 0u64);
        self.check(unsafe { kmc_count_file(self.ctx, c.as_ptr(), &mut nd, &mut nt) })?;
        Ok((nd, nt))
    }

    /// Sorted table: (key as ASCII, count), ascending == the order of `lr_chunk.sort()` (main.rs:87).
    pub fn table(&mut self) -> Result<Vec<(String, u64)>, KmcError> {
        let (mut nd, mut nt) = (0u64, 0u64);
        self.check(unsafe { kmc_finalize(self.ctx, &mut nd, &mut nt) })?;
        let n = nd as usize;
        let (mut hi, mut lo, mut cnt) = (vec![0u64; n], vec![0u64; n], vec![0u64; n]);
        self.check(unsafe { kmc_export(self.ctx, hi.as_mut_ptr(), lo.as_mut_ptr(), cnt.as_mut_ptr(), nd) })?;
        let mut buf = vec![0u8; self.klen as usize];
        let mut out = Vec::with_capacity(n);
        for i in 0..n {
            unsafe { kmc_decode_key(hi[i], lo[i], self.klen, buf.as_mut_ptr() as *mut c_char) };
            out.push((String::from_utf8_lossy(&buf).into_owned(), cnt[i]));
        }
        Ok(out)
    }
}

impl Drop for Counter {
    fn drop(&mut self) {
        unsafe { kmc_destroy(self.ctx) }
    }
}
