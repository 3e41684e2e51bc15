//! Rust binding of `include/kmc.h` (libkmc.so).  NOT COMPILED in the build environment of this
//! repository (no rustc there); kept in sync with the header by hand.
use std::ffi::CStr;
use std::os::raw::{c_char, c_int, c_void};

pub const KMC_MODE_CONTIG: i32 = 0;
pub const KMC_MODE_LR: i32 = 1;
pub const KMC_ALGO_AUTO: i32 = 0;

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
pub struct KmcCtx {
    _private: [u8; 0],
}

extern "C" {
    pub fn kmc_create(out: *mut *mut KmcCtx, cfg: *const KmcConfig) -> c_int;
    pub fn kmc_destroy(ctx: *mut KmcCtx);
    pub fn kmc_last_error(ctx: *const KmcCtx) -> *const c_char;
    pub fn kmc_reset(ctx: *mut KmcCtx) -> c_int;
    pub fn kmc_add_batch(ctx: *mut KmcCtx, bases: *const u8, offsets: *const u64, n_reads: u64) -> c_int;
    pub fn kmc_finalize(ctx: *mut KmcCtx, n_distinct: *mut u64, n_total: *mut u64) -> c_int;
    pub fn kmc_export(ctx: *mut KmcCtx, key_hi: *mut u64, key_lo: *mut u64, count: *mut u64, cap: u64) -> c_int;
    pub fn kmc_decode_key(key_hi: u64, key_lo: u64, klen: c_int, out: *mut c_char);
    pub fn kmc_count_file(ctx: *mut KmcCtx, path: *const c_char, n_distinct: *mut u64, n_total: *mut u64) -> c_int;
    pub fn kmc_add_batch_device(ctx: *mut KmcCtx, d_bases: *const c_void, d_offsets: *const c_void, n_reads: u64, n_bases: u64, max_read_len: u64) -> c_int;
    pub fn kmc_poll(ctx: *mut KmcCtx) -> c_int;
    pub fn kmc_forget_source(ctx: *mut KmcCtx, what: c_int) -> c_int;
    // multi-GPU reduce (one process per GPU; the collective itself is the host program's, e.g. RCCL)
    pub fn kmc_slab_words(ctx: *const KmcCtx, slab_entries: u64) -> u64;
    pub fn kmc_pack_slab_device(ctx: *mut KmcCtx, d_slab: *mut c_void, slab_entries: u64) -> c_int;
    pub fn kmc_merge_slabs_device(ctx: *mut KmcCtx, d_slabs: *const c_void, n_slabs: u32, slab_entries: u64, my_part: u32, n_parts: u32) -> c_int;
    pub fn kmc_partition_device(ctx: *mut KmcCtx, n_parts: u32, part_begin: *mut u64, d_key_hi: *mut *const c_void,
                                d_key_lo: *mut *const c_void, d_count: *mut *const c_void) -> c_int;
    pub fn kmc_merge_pairs_device(ctx: *mut KmcCtx, d_key_hi: *const c_void, d_key_lo: *const c_void, d_count: *const c_void, n_pairs: u64) -> c_int;
    // streaming host reader (chunks end at record boundaries; buffers owned by the stream)
    pub fn kmc_fasta_stream_open(path: *const c_char, chunk_bytes: u64, out: *mut *mut KmcFastaStream, errbuf: *mut c_char, errbuf_len: usize) -> c_int;
    pub fn kmc_fasta_stream_next(s: *mut KmcFastaStream, out: *mut KmcReads, eof: *mut c_int, errbuf: *mut c_char, errbuf_len: usize) -> c_int;
    pub fn kmc_fasta_stream_close(s: *mut KmcFastaStream);
}

#[repr(C)]
pub struct KmcFastaStream {
    _private: [u8; 0],
}

#[repr(C)]
pub struct KmcReads {
    pub bases: *mut u8,
    pub offsets: *mut u64,
    pub n_reads: u64,
    pub n_bases: u64,
    pub max_read_len: u64,
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
