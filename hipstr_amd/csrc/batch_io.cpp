// batch_io.cpp — flat on-disk / wire form of a hipstr_batch_t: what a CPU worker that decoded and QC'd the
// reads of a region shard hands to the process that owns a GPU.  Host only; no device is touched.
//
// Layout (little endian): 64-byte header { magic "HSTRBAT1", u32 version, u32 n_loci, u64 n_sections, u64 total_bytes, u64
// fnv1a-64 of everything after the header, 24 reserved }, then n_sections x { u32 id, u32 elem_bytes, u64 count }, then the
// section payloads in the same order, each padded to 8 bytes.  Sections are the arrays of hipstr_batch_t by field order; absent
// optional arrays (realign_hap, realign_read) have count 0.  A reader maps the file into one allocation and points a
// hipstr_batch_t into it — no per-array copies.
#include <climits>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/hipstr_hmm.h"
#include "api_internal.h"

namespace {

const char kMagic[8] = { 'H', 'S', 'T', 'R', 'B', 'A', 'T', '1' };
const uint32_t kVersion = 1;
enum { N_SECTIONS = 18 };

struct Header { char magic[8]; uint32_t version, n_loci; uint64_t n_sections, total_bytes, checksum; uint8_t reserved[24]; };
struct Section { uint32_t id, elem_bytes; uint64_t count; };
static_assert(sizeof(Header) == 64 && sizeof(Section) == 16, "on-disk structs");

uint64_t fnv1a(const uint8_t* p, size_t n){
  uint64_t h = 1469598103934665603ull;
  for (size_t i = 0; i < n; i++){ h ^= p[i]; h *= 1099511628211ull; }
  return h;
}

struct View { const void* ptr; uint32_t elem; uint64_t count; };

// sizes of every array of a batch, derived from its own offset arrays
int sections_of(const hipstr_batch_t* b, View v[N_SECTIONS], std::string& err){
  const int64_t n = b->n_loci;
  if (n < 0){ err = "negative locus count"; return 1; }
  int64_t n_opts = 0;
  for (int64_t i = 0; i < 3*n; i++){ if (b->blk_nopts[i] < 1){ err = "haplotype block without options"; return 1; } n_opts += b->blk_nopts[i]; }
  const int64_t n_haps = n ? b->hap_off[n] : 0, n_reads = n ? b->read_off[n] : 0;
  const int64_t n_seq = n_opts ? b->opt_off[n_opts] : 0, n_bases = n_reads ? b->base_off[n_reads] : 0, n_cig = n_reads ? b->cigar_off[n_reads] : 0;
  const View t[N_SECTIONS] = {
    { b->blk_start, 4, (uint64_t)(3*n) }, { b->blk_end, 4, (uint64_t)(3*n) }, { b->blk_nopts, 4, (uint64_t)(3*n) }, { b->period, 4, (uint64_t)n },
    { b->stutter, 8, (uint64_t)(6*n) }, { b->opt_off, 4, (uint64_t)(n_opts + 1) }, { b->seq, 1, (uint64_t)n_seq }, { b->hap_off, 4, (uint64_t)(n + 1) },
    { b->realign_hap, 1, b->realign_hap ? (uint64_t)n_haps : 0 }, { b->read_off, 4, (uint64_t)(n + 1) }, { b->base_off, 4, (uint64_t)(n_reads + 1) },
    { b->bases, 1, (uint64_t)n_bases }, { b->quals, 1, (uint64_t)n_bases }, { b->read_start, 4, (uint64_t)n_reads },
    { b->cigar_off, 4, (uint64_t)(n_reads + 1) }, { b->cigar_op, 1, (uint64_t)n_cig }, { b->cigar_len, 4, (uint64_t)n_cig },
    { b->realign_read, 1, b->realign_read ? (uint64_t)n_reads : 0 } };
  memcpy(v, t, sizeof t);
  return 0;
}

// offsets[0..n] must start at 0, be non-decreasing and end at `total`
bool offsets_ok(const int32_t* off, uint64_t n, uint64_t total){
  if (off[0] != 0) return false;
  for (uint64_t i = 0; i < n; i++) if (off[i+1] < off[i]) return false;
  return (uint64_t)(int64_t)off[n] == total;
}

// Consistency of a deserialized image, using only the section counts as trusted sizes (sec[] itself was bounds-checked against
// the image).  Order matters: an array is read only after its own count has been checked.
int validate_image(const hipstr_batch_t& b, const Section* sec, std::string& err){
  const int64_t n = b.n_loci;
  if (n < 0){ err = "negative locus count"; return 1; }
  const uint64_t un = (uint64_t)n;
  static const int per_locus3[] = { 0, 1, 2 };
  for (int s : per_locus3) if (sec[s].count != 3*un){ err = "block table size"; return 1; }
  if (sec[3].count != un || sec[4].count != 6*un){ err = "period / stutter size"; return 1; }
  if (n == 0){
    for (int s = 5; s < N_SECTIONS; s++) if (sec[s].count > 1){ err = "data without loci"; return 1; }
    return 0;
  }
  if (sec[7].count != un + 1 || sec[9].count != un + 1){ err = "hap_off / read_off size"; return 1; }
  uint64_t n_opts = 0;
  for (uint64_t i = 0; i < 3*un; i++){
    if (b.blk_nopts[i] < 1){ err = "haplotype block without options"; return 1; }
    n_opts += (uint64_t)b.blk_nopts[i];
  }
  if (sec[5].count != n_opts + 1){ err = "opt_off size"; return 1; }
  if (!offsets_ok(b.opt_off, n_opts, sec[6].count)){ err = "opt_off is not a prefix sum of seq"; return 1; }
  // hap_off: prefix sums of num_combs = product of the block option counts
  if (b.hap_off[0] != 0){ err = "hap_off"; return 1; }
  for (uint64_t l = 0; l < un; l++){
    const int64_t combs = (int64_t)b.blk_nopts[3*l]*b.blk_nopts[3*l+1]*b.blk_nopts[3*l+2];
    if (combs > INT32_MAX || (int64_t)b.hap_off[l+1] - b.hap_off[l] != combs){ err = "hap_off is not the prefix sum of num_combs"; return 1; }
    if (b.blk_end[3*l] < b.blk_start[3*l] || b.blk_end[3*l+1] < b.blk_start[3*l+1] || b.blk_end[3*l+2] < b.blk_start[3*l+2]){ err = "block coordinates"; return 1; }
  }
  const uint64_t n_haps = (uint64_t)b.hap_off[un];
  if (sec[8].count != 0 && sec[8].count != n_haps){ err = "realign_hap size"; return 1; }
  if (b.read_off[0] != 0){ err = "read_off"; return 1; }
  for (uint64_t l = 0; l < un; l++) if (b.read_off[l+1] < b.read_off[l]){ err = "read_off decreases"; return 1; }
  const uint64_t n_reads = (uint64_t)b.read_off[un];
  if (sec[10].count != n_reads + 1 || sec[14].count != n_reads + 1 || sec[13].count != n_reads){ err = "per-read table size"; return 1; }
  if (sec[17].count != 0 && sec[17].count != n_reads){ err = "realign_read size"; return 1; }
  if (sec[11].count != sec[12].count || sec[15].count != sec[16].count){ err = "bases/quals or cigar op/len size"; return 1; }
  if (!offsets_ok(b.base_off, n_reads, sec[11].count)){ err = "base_off is not a prefix sum of bases"; return 1; }
  if (!offsets_ok(b.cigar_off, n_reads, sec[15].count)){ err = "cigar_off is not a prefix sum of the CIGAR pool"; return 1; }
  return 0;
}

}  // namespace

struct hipstr_batch_file { std::vector<uint8_t> mem; hipstr_batch_t batch; };

extern "C" {

int64_t hipstr_batch_serialized_size(const hipstr_batch_t* b){
  View v[N_SECTIONS]; std::string err;
  if (!b || sections_of(b, v, err)){ hipstr::api_fail(b ? err : "null argument"); return -1; }
  uint64_t bytes = sizeof(Header) + N_SECTIONS*sizeof(Section);
  for (int s = 0; s < N_SECTIONS; s++) bytes += (v[s].elem*v[s].count + 7) & ~(uint64_t)7;
  return (int64_t)bytes;
}

int hipstr_batch_serialize(const hipstr_batch_t* b, void* out, int64_t cap){
  const int64_t need = hipstr_batch_serialized_size(b);
  if (need < 0) return 1;
  if (!out || cap < need) return hipstr::api_fail("serialization buffer is too small");
  View v[N_SECTIONS]; std::string err;
  sections_of(b, v, err);
  uint8_t* p = (uint8_t*)out;
  memset(p, 0, (size_t)need);
  Header h; memset(&h, 0, sizeof h);
  memcpy(h.magic, kMagic, 8); h.version = kVersion; h.n_loci = (uint32_t)b->n_loci; h.n_sections = N_SECTIONS; h.total_bytes = (uint64_t)need;
  Section* sec = (Section*)(p + sizeof(Header));
  uint8_t* q = p + sizeof(Header) + N_SECTIONS*sizeof(Section);
  for (int s = 0; s < N_SECTIONS; s++){
    sec[s].id = (uint32_t)s; sec[s].elem_bytes = v[s].elem; sec[s].count = v[s].count;
    const size_t bytes = (size_t)(v[s].elem*v[s].count);
    if (bytes) memcpy(q, v[s].ptr, bytes);
    q += (bytes + 7) & ~(size_t)7;
  }
  h.checksum = fnv1a(p + sizeof(Header), (size_t)need - sizeof(Header));
  memcpy(p, &h, sizeof h);
  return 0;
}

// Takes ownership of a copy of the bytes; the returned batch points into it.
hipstr_batch_file_t* hipstr_batch_deserialize(const void* data, int64_t size){
  if (!data || size < (int64_t)(sizeof(Header) + N_SECTIONS*sizeof(Section))){ hipstr::api_fail("not a hipstr batch: too short"); return NULL; }
  Header h; memcpy(&h, data, sizeof h);
  if (memcmp(h.magic, kMagic, 8) != 0){ hipstr::api_fail("not a hipstr batch: bad magic"); return NULL; }
  if (h.version != kVersion || h.n_sections != N_SECTIONS){ hipstr::api_fail("unsupported hipstr batch version"); return NULL; }
  if ((int64_t)h.total_bytes != size){ hipstr::api_fail("hipstr batch is truncated or has trailing bytes"); return NULL; }
  if (fnv1a((const uint8_t*)data + sizeof(Header), (size_t)size - sizeof(Header)) != h.checksum){ hipstr::api_fail("hipstr batch checksum mismatch"); return NULL; }
  hipstr_batch_file_t* f = new hipstr_batch_file_t();
  f->mem.assign((const uint8_t*)data, (const uint8_t*)data + size);
  const Section* sec = (const Section*)(f->mem.data() + sizeof(Header));
  const uint8_t* q = f->mem.data() + sizeof(Header) + N_SECTIONS*sizeof(Section);
  const void* ptr[N_SECTIONS];
  static const uint32_t elem[N_SECTIONS] = { 4, 4, 4, 4, 8, 4, 1, 4, 1, 4, 4, 1, 1, 4, 4, 1, 4, 1 };
  for (int s = 0; s < N_SECTIONS; s++){
    const uint64_t bytes = (uint64_t)sec[s].elem_bytes*sec[s].count;
    if (sec[s].id != (uint32_t)s || sec[s].elem_bytes != elem[s] || (uint64_t)(q - f->mem.data()) + bytes > (uint64_t)size){
      delete f; hipstr::api_fail("hipstr batch section table is inconsistent"); return NULL;
    }
    ptr[s] = sec[s].count ? (const void*)q : NULL;
    q += (bytes + 7) & ~(uint64_t)7;
  }
  hipstr_batch_t& b = f->batch;
  b.n_loci = (int32_t)h.n_loci;
  b.blk_start = (const int32_t*)ptr[0]; b.blk_end = (const int32_t*)ptr[1]; b.blk_nopts = (const int32_t*)ptr[2]; b.period = (const int32_t*)ptr[3];
  b.stutter = (const double*)ptr[4]; b.opt_off = (const int32_t*)ptr[5]; b.seq = (const char*)ptr[6]; b.hap_off = (const int32_t*)ptr[7];
  b.realign_hap = (const uint8_t*)ptr[8]; b.read_off = (const int32_t*)ptr[9]; b.base_off = (const int32_t*)ptr[10];
  b.bases = (const char*)ptr[11]; b.quals = (const char*)ptr[12]; b.read_start = (const int32_t*)ptr[13]; b.cigar_off = (const int32_t*)ptr[14];
  b.cigar_op = (const char*)ptr[15]; b.cigar_len = (const int32_t*)ptr[16]; b.realign_read = (const uint8_t*)ptr[17];
  // The arrays must describe each other.  The image is untrusted: every array is checked against the section counts BEFORE an
  // entry of it is used as an index or a size, and every offset array must start at 0, never decrease and end at the count of
  // the section it indexes — so nothing downstream (prepare_batch) can be driven to a negative length or out of the image.
  std::string err;
  if (validate_image(b, sec, err)){ delete f; hipstr::api_fail("hipstr batch arrays do not match their offset tables: " + err); return NULL; }
  return f;
}

const hipstr_batch_t* hipstr_batch_file_batch(const hipstr_batch_file_t* f){ return f ? &f->batch : NULL; }
void hipstr_batch_file_free(hipstr_batch_file_t* f){ delete f; }

int hipstr_batch_write(const char* path, const hipstr_batch_t* b){
  const int64_t need = hipstr_batch_serialized_size(b);
  if (need < 0) return 1;
  std::vector<uint8_t> buf((size_t)need);
  if (hipstr_batch_serialize(b, buf.data(), need)) return 1;
  FILE* fp = fopen(path, "wb");
  if (!fp) return hipstr::api_fail(std::string("cannot open for writing: ") + path);
  const bool ok = fwrite(buf.data(), 1, buf.size(), fp) == buf.size();
  if (fclose(fp) != 0 || !ok) return hipstr::api_fail(std::string("short write: ") + path);
  return 0;
}

hipstr_batch_file_t* hipstr_batch_read(const char* path){
  FILE* fp = fopen(path, "rb");
  if (!fp){ hipstr::api_fail(std::string("cannot open: ") + path); return NULL; }
  std::vector<uint8_t> buf;
  uint8_t tmp[1 << 16]; size_t got;
  while ((got = fread(tmp, 1, sizeof tmp, fp)) > 0) buf.insert(buf.end(), tmp, tmp + got);
  fclose(fp);
  return hipstr_batch_deserialize(buf.data(), (int64_t)buf.size());
}

}  // extern "C"
