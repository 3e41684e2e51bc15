//! The reference's `main()` (k-mer-count/src/main.rs:43-91) with its loop nest (:63-87) handed to
//! the GPU.  NOT COMPILED in the build environment of this repository (no rustc there).
//!
//!   k-mer-count [FASTA] [-k K] [--forward]
//!
//! No `-k`: the reference's own LR-gapped computation, one line per occurrence (main.rs:88-90).
//! `-k K`: contiguous canonical K-mers, `KMER\tCOUNT` lines.
use bio::io::fasta::{FastaRead, Reader, Record};
use k_mer_count::Counter;
use std::fs::File;

fn main() {
    let args: Vec<String> = std::env::args().collect();
    let mut path = "sample.fasta".to_string(); // main.rs:44
    let mut k: Option<i32> = None;
    let mut canonical = true;
    let mut i = 1;
    while i < args.len() {
        match args[i].as_str() {
            "-k" => { i += 1; k = Some(args[i].parse().expect("-k needs a number")); }
            "--forward" => canonical = false,
            p => path = p.to_string(),
        }
        i += 1;
    }
    let file = File::open(&path).expect("Error during opening the file"); // main.rs:44
    let mut reader = Reader::new(file); // main.rs:45
    let mut record = Record::new(); // main.rs:46

    let mut bases: Vec<u8> = Vec::new();
    let mut offsets: Vec<u64> = vec![0];
    loop {
        reader.read(&mut record).unwrap(); // main.rs:59
        if record.is_empty() {
            break; // main.rs:60-62
        }
        // No alphabet pre-check here: the reference panics only on a byte its bucket_sort inspects
        // (main.rs:17-23), i.e. a byte that some emitted chunk covers.  include/kmc.h states the same rule
        // for KMC_MODE_LR and libkmc applies it on the GPU: kmc_add_batch / kmc_finalize return
        // KMC_ERR_ALPHABET ("Unexpected charactor ...") for exactly those bytes, which the expect() calls
        // below turn into the reference's panic (exit code 101).  Bytes no window reads are accepted, as in
        // the reference and in the C++ CLI (kmc_cli.cpp).
        bases.extend_from_slice(record.seq());
        offsets.push(bases.len() as u64);
    }

    let mut counter = Counter::new(k, canonical, 0).expect("kmc_create");
    counter.add_batch(&bases, &offsets).expect("kmc_add_batch"); // replaces main.rs:63-81
    let table = counter.table().expect("kmc_finalize/kmc_export"); // replaces main.rs:84,87
    for (key, count) in table.iter() {
        if k.is_none() {
            for _ in 0..*count {
                println!("{}", key); // main.rs:88-90
            }
        } else {
            println!("{}\t{}", key, count);
        }
    }
}
