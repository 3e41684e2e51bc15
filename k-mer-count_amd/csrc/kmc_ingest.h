// kmc_ingest.h -- streaming host FASTA reader (internal to libkmc; the public form is
// kmc_fasta_stream_* in include/kmc.h).
//
// The file is mapped and cut into chunks at record starts; a chunk is parsed by worker threads
// (each takes a byte segment snapped to line starts: one memchr pass, header lines open records,
// other lines are right-trimmed and copied), every worker writing its sequence bytes at the same
// relative position its text has in the chunk -- output never exceeds input, so workers need no
// coordination and nothing is stitched on the host.  kmc_count_file uploads the workers' pieces
// straight to their final (dense) place in the device buffer and parses the next chunk while the
// GPU copies and counts this one.  Semantics are those of kmc_parse_fasta (the reader the
// reference uses, k-mer-count/src/main.rs:45-46,59-62).  Extension the reference does not have
// (parity unpinned): a file whose first byte is '@' is read as four-line FASTQ --
// header, sequence, '+' line, quality -- of which only the sequence line reaches the GPU.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <string>
#include <vector>

struct KmcIngestPiece {
    uint64_t src_off;   // where the worker wrote its bytes, relative to the chunk's output buffer
    uint64_t n_bytes;   // sequence bytes of this piece
    uint64_t dst_off;   // their place in the dense concatenation of the chunk
};

struct KmcIngestChunk {
    std::vector<KmcIngestPiece> pieces;
    std::vector<uint64_t> offsets;  // n_reads + 1, dense coordinates
    uint64_t n_reads = 0, n_bases = 0, max_read_len = 0;
    bool eof = false;               // nothing follows this chunk
    bool terminated = false;        // an empty record ended the input inside this chunk (nothing after it counts)
    int bad_byte = -1;              // first byte outside ACGT (only looked for when asked)
};

class KmcFastaIngest {
  public:
    ~KmcFastaIngest();
    // KMC_OK, or KMC_ERR_IO (cannot open / map); chunk_bytes: text bytes per chunk (a chunk ends at the
    // first record start at or after that many bytes)
    int open(const char* path, uint64_t chunk_bytes, std::string* err);
    uint64_t chunk_capacity() const;  // bytes a chunk's output buffer must hold (longest chunk of this file)
    // Parse the next chunk into `out_buf` (chunk_capacity() bytes).  KMC_OK (chunk->eof tells whether more
    // follows; a chunk may hold zero reads), KMC_ERR_FORMAT ("Expected > at record start.").
    int next(uint8_t* out_buf, bool check_alphabet, KmcIngestChunk* chunk, std::string* err);
    // The same for chunk `idx` (0 .. n_chunks()-1) without touching the reader's position: feeder threads of a
    // multi-GPU run parse different chunks of one file concurrently, `threads` parser threads each.
    size_t n_chunks() const;
    unsigned threads() const { return threads_; }
    int parse_chunk(size_t idx, unsigned threads, uint8_t* out_buf, bool check_alphabet, KmcIngestChunk* chunk, std::string* err) const;

  private:
    const char* map_ = nullptr;
    uint64_t size_ = 0, pos_ = 0, chunk_bytes_ = 0, cap_ = 0;
    std::vector<uint64_t> cuts_;  // chunk boundaries (record starts), cuts_[0] = 0 ... cuts_.back() = size_
    size_t next_cut_ = 0;
    bool done_ = false;
    bool fastq_ = false;  // the file starts with '@': four-line FASTQ records (header, sequence, '+', quality)
    unsigned threads_ = 1;
    uint64_t fastq_record_start(uint64_t q, uint64_t limit) const;
};
