// lcd_io.cpp -- SURVEY 8(f) f3: the data formats in front of the hot path, without htslib (absent from the reference checkout).
//   * BGZF container + BAM records: what collect_ref_seq_bam_main (src/bam_utils.c:1659-1716) loads for one region -- primary, mapped reads of
//     MAPQ >= min_mq overlapping [reg_beg, reg_end], in file order -- flattened to exactly the arrays lcd_digar_batch and lcd_read_view_t take
//     (0-based position, CIGAR words, BAM 4-bit bases, qualities).  BGZF blocks are independent deflate streams: block boundaries come from the
//     BSIZE fields, the blocks are inflated on host threads in parallel (zlib), records are decoded from the concatenation.  lcd_bam_load_region scans the
//     file; lcd_bam_load_region_indexed reads only the blocks the .bai points to (bins + linear index, SAM specification 5.2-5.3); both give what
//     sam_itr_queryi on (reg_beg - 1, reg_end] gives.
//   * FASTA + .fai: faidx_fetch_seq of a region as byte codes (nst_nt4_table).
//   * the VCF header lines of write_vcf_header (src/vcf_utils.c:17-96).  htslib's bcf_hdr_write decides their final order and adds
//     ##fileformat itself; that library is absent, so the header text is this project's rendering (the BODY lines are lcd_format_vcf's, exact).
#include <hip/hip_runtime.h>
#include <zlib.h>
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include "../../include/lcd_hotpath.h"
#include "lcd_io_internal.h"

namespace {
thread_local std::string g_io_err;
int io_err(int code, const std::string &m) { g_io_err = m; return code; }

int read_file(const char *path, std::vector<uint8_t> &buf) {
    FILE *f = fopen(path, "rb");
    if (!f) return io_err(-30, std::string("cannot open ") + path);
    fseek(f, 0, SEEK_END); const long n = ftell(f); fseek(f, 0, SEEK_SET);
    buf.resize(n > 0 ? (size_t)n : 0);
    const size_t got = n > 0 ? fread(buf.data(), 1, (size_t)n, f) : 0;
    fclose(f);
    return got == buf.size() ? 0 : io_err(-30, std::string("short read on ") + path);
}
struct Blk { size_t cdata, clen, uoff, ulen; uint32_t crc; };
// a view of the file image (the loaders hold it in a vector, lcd_bgzf_inflate_dev gets the caller's bytes)
struct Bytes { const uint8_t *p; size_t n; size_t size() const { return n; } const uint8_t &operator[](size_t i) const { return p[i]; } };
// BGZF: gzip members with an extra field 'B','C',2,BSIZE (total block size - 1); CRC32 and ISIZE (uncompressed length) are the last 8 bytes
int bgzf_blocks(const Bytes f, std::vector<Blk> &blks, size_t *total) {
    size_t o = 0, u = 0;
    while (o + 18 <= f.size()) {
        if (f[o] != 31 || f[o + 1] != 139 || f[o + 2] != 8 || !(f[o + 3] & 4)) return io_err(-31, "not a BGZF block");
        const unsigned xlen = f[o + 10] | (f[o + 11] << 8);
        if (o + 12 + (size_t)xlen + 8 > f.size()) return io_err(-31, "BGZF block: extra field runs past the end of the file");
        size_t x = o + 12; unsigned bsize = 0; bool found = false;
        while (x + 4 <= o + 12 + xlen) {
            const unsigned slen = f[x + 2] | (f[x + 3] << 8);
            if (f[x] == 'B' && f[x + 1] == 'C' && slen == 2) { bsize = f[x + 4] | (f[x + 5] << 8); found = true; }
            x += 4 + slen;
        }
        if (!found || o + bsize + 1 > f.size()) return io_err(-31, "BGZF block without BSIZE / truncated file");
        const size_t end = o + bsize + 1;
        if (end < o + 12 + (size_t)xlen + 8) return io_err(-31, "malformed BGZF block: BSIZE smaller than header + trailer"); // (before any f[end - k] is read: a hostile BSIZE must not index in front of the block)
        const unsigned isize = f[end - 4] | (f[end - 3] << 8) | (f[end - 2] << 16) | ((unsigned)f[end - 1] << 24);
        Blk b; b.cdata = o + 12 + xlen; b.clen = end - 8 - b.cdata; b.uoff = u; b.ulen = isize;
        b.crc = f[end - 8] | (f[end - 7] << 8) | (f[end - 6] << 16) | ((uint32_t)f[end - 5] << 24);
        if (end < 8 + b.cdata || isize > 65536) return io_err(-31, "malformed BGZF block");
        if (isize) blks.push_back(b);
        u += isize; o = end;
    }
    *total = u;
    return 0;
}
int bgzf_inflate_all(const std::vector<uint8_t> &f, std::vector<uint8_t> &out, int n_threads) {
    std::vector<Blk> blks; size_t total = 0;
    if (int rc = bgzf_blocks(Bytes{f.data(), f.size()}, blks, &total)) return rc;
    out.resize(total);
    std::atomic<size_t> next{0}; std::atomic<int> bad{0};
    auto work = [&]() {
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= blks.size()) return;
            z_stream zs; memset(&zs, 0, sizeof(zs));
            if (inflateInit2(&zs, -15) != Z_OK) { bad = 1; return; }
            zs.next_in = (Bytef *)(f.data() + blks[i].cdata); zs.avail_in = (uInt)blks[i].clen;
            zs.next_out = out.data() + blks[i].uoff; zs.avail_out = (uInt)blks[i].ulen;
            const int r = inflate(&zs, Z_FINISH);
            if (r != Z_STREAM_END || zs.total_out != blks[i].ulen) bad = 1;
            inflateEnd(&zs);
        }
    };
    n_threads = std::max(1, std::min<int>(n_threads, (int)blks.size()));
    std::vector<std::thread> ths;
    for (int t = 1; t < n_threads; ++t) ths.emplace_back(work);
    work();
    for (auto &t : ths) t.join();
    return bad ? io_err(-32, "inflate failed on a BGZF block") : 0;
}
inline int32_t le32(const uint8_t *p) { return (int32_t)(p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24)); }
inline uint8_t nt4c(unsigned char c) { return (c == 'A' || c == 'a') ? 0 : (c == 'C' || c == 'c') ? 1 : (c == 'G' || c == 'g') ? 2 : (c == 'T' || c == 't') ? 3 : 4; }
template <typename T> T *dup_vec(const std::vector<T> &v) { T *p = (T *)malloc((v.size() + 1) * sizeof(T)); if (!v.empty()) memcpy(p, v.data(), v.size() * sizeof(T)); return p; }

// one region's records, filtered as collect_ref_seq_bam_main does (src/bam_utils.c:1672-1706), flattened; shared by the scan and the indexed loader
struct Collector {
    const int tid; const int64_t reg_beg, reg_end; const int min_mapq;
    std::vector<int64_t> pos0, endp; std::vector<int> mapq, flag, ncig, qlen;
    std::vector<uint64_t> coff, soff, qoff, noff; std::vector<uint32_t> cpool; std::vector<uint8_t> spool, qpool; std::vector<char> npool;
    Collector(int t, int64_t b, int64_t e, int mq) : tid(t), reg_beg(b), reg_end(e), min_mapq(mq) {}
    // the real CIGAR of a read with more than 65 535 operations (ultra-long ONT reads): BAM keeps the placeholder `<l_seq>S<ref_len>N` in the 16-bit field and the
    // operations in the CG:B,I tag; htslib's bam_read1 (behind the reference's sam_itr_next) swaps them in.  Returns the tag's operations, or nullptr.
    static const uint8_t *cg_tag(const uint8_t *aux, const uint8_t *end, uint32_t *n, const uint32_t n_cigar) {
        while (aux + 3 <= end) {
            const uint8_t t0 = aux[0], t1 = aux[1], ty = aux[2]; aux += 3;
            size_t sz;
            if (t0 == 'C' && t1 == 'G' && ty != 'B') return nullptr; // (bam_aux_get: the first CG tag decides; another type is not a CIGAR)
            switch (ty) {
                case 'A': case 'c': case 'C': sz = 1; break;
                case 's': case 'S': sz = 2; break;
                case 'i': case 'I': case 'f': sz = 4; break;
                case 'Z': case 'H': { const uint8_t *q = aux; while (q < end && *q) ++q; if (q >= end) return nullptr; sz = (size_t)(q - aux) + 1; break; }
                case 'B': {
                    if (aux + 5 > end) return nullptr;
                    const uint8_t sub = aux[0]; const uint32_t cnt = (uint32_t)le32(aux + 1);
                    const size_t es = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : (sub == 'i' || sub == 'I' || sub == 'f') ? 4 : 0;
                    if (!es || (size_t)(end - (aux + 5)) < (size_t)cnt * es) return nullptr;
                    if (t0 == 'C' && t1 == 'G') { if ((sub == 'I' || sub == 'i') && cnt >= n_cigar && cnt < (1u << 29)) { *n = cnt; return aux + 5; } return nullptr; }
                    sz = 5 + (size_t)cnt * es; break;
                }
                default: return nullptr;
            }
            if ((size_t)(end - aux) < sz) return nullptr;
            aux += sz;
        }
        return nullptr;
    }
    // r: the record behind its block_size word, bs bytes long.  1 taken, 0 skipped, -1 nothing further can overlap (sorted input), -2 malformed (a field runs past
    // the record, or a placeholder CIGAR without its CG tag)
    int take(const uint8_t *r, const size_t bs) {
        const int refid = le32(r), p = le32(r + 4), lname = r[8], mq = r[9], fl = r[14] | (r[15] << 8), lseq = le32(r + 16);
        int nc = r[12] | (r[13] << 8);
        if (lseq < 0 || 32 + (size_t)lname + 4 * (size_t)nc + ((size_t)lseq + 1) / 2 + (size_t)lseq > bs) return -2;
        if (refid != tid) return (refid > tid || refid < 0) && !pos0.empty() ? -1 : 0;
        const uint8_t *cg = r + 32 + lname;
        const uint8_t *const sq = cg + 4 * (size_t)nc, *const ql = sq + (lseq + 1) / 2;
        // htslib's bam_tag2cigar: a mapped record whose first operation is `<l_seq>S` takes its operations from a CG:B,I / B,i tag with >= n_cigar (and < 2^29)
        // entries; without such a tag the record keeps its own CIGAR (no error: htslib returns 0 and goes on)
        if (nc >= 1 && refid >= 0 && p >= 0) {
            const uint32_t c0 = (uint32_t)le32(cg);
            if ((c0 & 0xf) == 4 && (int)(c0 >> 4) == lseq) {
                uint32_t n = 0;
                const uint8_t *real = cg_tag(ql + lseq, r + bs, &n, (uint32_t)nc);
                if (real) { cg = real; nc = (int)n; }
            }
        }
        int64_t rl = 0;
        for (int k = 0; k < nc; ++k) { const uint32_t c = (uint32_t)le32(cg + 4 * k); const int op = c & 0xf; if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rl += c >> 4; }
        const int64_t e0 = p + (rl > 0 ? rl : 1); // bam_endpos: 0-based exclusive end (an alignment without reference bases spans one)
        if (p >= reg_end) return -1;               // sorted input: nothing further overlaps (reg_beg - 1, reg_end]
        if (e0 <= reg_beg - 1) return 0;
        if ((fl & (0x4 | 0x100 | 0x800)) || mq < min_mapq) return 0; // BAM_FUNMAP | BAM_FSECONDARY | BAM_FSUPPLEMENTARY, src/bam_utils.c:1683
        pos0.push_back(p); endp.push_back(e0); mapq.push_back(mq); flag.push_back(fl); ncig.push_back(nc); qlen.push_back(lseq);
        coff.push_back(cpool.size()); for (int k = 0; k < nc; ++k) cpool.push_back((uint32_t)le32(cg + 4 * k));
        soff.push_back(spool.size()); spool.insert(spool.end(), sq, sq + (lseq + 1) / 2);
        qoff.push_back(qpool.size()); qpool.insert(qpool.end(), ql, ql + lseq);
        noff.push_back(npool.size()); npool.insert(npool.end(), (const char *)r + 32, (const char *)r + 32 + lname);
        return 1;
    }
    int finish(lcd_bam_reads_t *out, int tid_, int64_t tlen, int n_ref) {
        out->n_reads = (int)pos0.size(); out->tid = tid_; out->target_len = tlen; out->n_targets = n_ref;
        out->pos0 = dup_vec(pos0); out->end_pos = dup_vec(endp); out->mapq = dup_vec(mapq); out->flag = dup_vec(flag); out->n_cigar = dup_vec(ncig); out->qlen = dup_vec(qlen);
        out->cigar_off = dup_vec(coff); out->cigar_pool = dup_vec(cpool); out->seq_off = dup_vec(soff); out->seq_pool = dup_vec(spool);
        out->qual_off = dup_vec(qoff); out->qual_pool = dup_vec(qpool); out->name_off = dup_vec(noff); out->name_pool = dup_vec(npool);
        return out->n_reads;
    }
};

// BGZF as a stream with virtual offsets (compressed offset of the block << 16 | offset inside the inflated block), for the indexed loader: seek to a chunk's
// start, read records across block boundaries, ask where the next byte is
struct BgzfStream {
    FILE *f = nullptr; uint64_t coff = 0, next_coff = 0; std::vector<uint8_t> blk; size_t at = 0; bool eof = false;
    ~BgzfStream() { if (f) fclose(f); }
    int load(uint64_t c) { // the block at compressed offset c
        uint8_t h[18];
        if (fseek(f, (long)c, SEEK_SET) != 0 || fread(h, 1, 18, f) != 18) { eof = true; blk.clear(); at = 0; coff = c; next_coff = c; return 0; }
        if (h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) return io_err(-31, "not a BGZF block");
        const unsigned xlen = h[10] | (h[11] << 8);
        std::vector<uint8_t> x(xlen);
        memcpy(x.data(), h + 12, std::min<size_t>(6, xlen));
        if (xlen > 6 && fread(x.data() + 6, 1, xlen - 6, f) != xlen - 6) return io_err(-31, "truncated BGZF header");
        unsigned bsize = 0; bool found = false;
        for (size_t q = 0; q + 4 <= xlen;) { const unsigned slen = x[q + 2] | (x[q + 3] << 8); if (x[q] == 'B' && x[q + 1] == 'C' && slen == 2 && q + 6 <= xlen) { bsize = x[q + 4] | (x[q + 5] << 8); found = true; } q += 4 + slen; }
        if (!found) return io_err(-31, "BGZF block without BSIZE");
        const size_t clen = (size_t)bsize + 1 - 12 - xlen; // deflate data + CRC32 + ISIZE
        std::vector<uint8_t> cd(clen);
        if (xlen < 6) { if (fseek(f, (long)(c + 12 + xlen), SEEK_SET) != 0) return io_err(-31, "seek failed"); }
        if (clen < 8 || fread(cd.data(), 1, clen, f) != clen) return io_err(-31, "truncated BGZF block");
        const unsigned isize = cd[clen - 4] | (cd[clen - 3] << 8) | (cd[clen - 2] << 16) | ((unsigned)cd[clen - 1] << 24);
        blk.resize(isize);
        if (isize) {
            z_stream zs; memset(&zs, 0, sizeof(zs));
            if (inflateInit2(&zs, -15) != Z_OK) return io_err(-32, "inflateInit2 failed");
            zs.next_in = cd.data(); zs.avail_in = (uInt)(clen - 8); zs.next_out = blk.data(); zs.avail_out = isize;
            const int r = inflate(&zs, Z_FINISH); const bool ok = r == Z_STREAM_END && zs.total_out == isize;
            inflateEnd(&zs);
            if (!ok) return io_err(-32, "inflate failed on a BGZF block");
        }
        coff = c; next_coff = c + bsize + 1; at = 0; eof = false;
        return 0;
    }
    int seek(uint64_t v) { if (int rc = load(v >> 16)) return rc; at = std::min<size_t>(v & 0xffff, blk.size()); return 0; }
    int settle() { while (!eof && at >= blk.size()) if (int rc = load(next_coff)) return rc; return 0; } // at a block's end the position is the next block's start
    uint64_t tell() { return (coff << 16) | (uint64_t)at; }
    int read(uint8_t *dst, size_t n) { // 0 ok, 1 end of file, < 0 error
        while (n) {
            if (int rc = settle()) return rc;
            if (eof) return 1;
            const size_t k = std::min(n, blk.size() - at);
            memcpy(dst, blk.data() + at, k); dst += k; at += k; n -= k;
        }
        return 0;
    }
};
// the bins a region may have records in (SAM specification 5.3, reg2bins; [beg, end) 0-based)
void reg2bins(int64_t beg, int64_t end, std::vector<uint32_t> &bins) {
    --end; bins.push_back(0);
    for (int64_t k = 1 + (beg >> 26); k <= 1 + (end >> 26); ++k) bins.push_back((uint32_t)k);
    for (int64_t k = 9 + (beg >> 23); k <= 9 + (end >> 23); ++k) bins.push_back((uint32_t)k);
    for (int64_t k = 73 + (beg >> 20); k <= 73 + (end >> 20); ++k) bins.push_back((uint32_t)k);
    for (int64_t k = 585 + (beg >> 17); k <= 585 + (end >> 17); ++k) bins.push_back((uint32_t)k);
    for (int64_t k = 4681 + (beg >> 14); k <= 4681 + (end >> 14); ++k) bins.push_back((uint32_t)k);
}
inline uint64_t le64(const uint8_t *p) { uint64_t v = 0; for (int i = 7; i >= 0; --i) v = (v << 8) | p[i]; return v; }
// header (-> tid, target length) and .bai lookup of a region: the merged chunks (virtual offsets, file order) whose records may overlap it
static int region_chunks(const char *bam_path, const char *bai_path, const char *chrom, int64_t reg_beg, int64_t reg_end, BgzfStream &bz, int &tid, int64_t &tlen, int &n_ref,
                         std::vector<std::pair<uint64_t, uint64_t>> &merged) {
    bz.f = fopen(bam_path, "rb");
    if (!bz.f) return io_err(-30, std::string("cannot open ") + bam_path);
    if (int rc = bz.seek(0)) return rc;
    uint8_t w[8];
    if (bz.read(w, 8) != 0 || memcmp(w, "BAM\1", 4) != 0) return io_err(-33, "not a BAM file");
    { std::vector<uint8_t> text((size_t)std::max(le32(w + 4), 0)); if (!text.empty() && bz.read(text.data(), text.size()) != 0) return io_err(-33, "truncated BAM header"); }
    if (bz.read(w, 4) != 0) return io_err(-33, "truncated BAM header");
    n_ref = le32(w);
    tid = -1; tlen = 0;
    for (int i = 0; i < n_ref; ++i) {
        if (bz.read(w, 4) != 0) return io_err(-33, "truncated BAM header");
        const int ln = le32(w);
        std::vector<uint8_t> nm((size_t)std::max(ln, 0) + 4);
        if (bz.read(nm.data(), (size_t)ln + 4) != 0) return io_err(-33, "truncated BAM header");
        if (std::string((const char *)nm.data(), (size_t)std::max(ln - 1, 0)) == chrom) { tid = i; tlen = le32(nm.data() + ln); }
    }
    if (tid < 0) return io_err(-34, std::string("contig not in the BAM header: ") + chrom);
    std::vector<uint8_t> ix;
    if (int rc = read_file(bai_path, ix)) return rc;
    if (ix.size() < 8 || memcmp(ix.data(), "BAI\1", 4) != 0 || le32(ix.data() + 4) <= tid) return io_err(-35, "not a .bai of this BAM");
    int64_t qb = reg_beg - 1, qe = reg_end; // 0-based half-open, as sam_itr_queryi(idx, tid, reg_beg - 1, reg_end)
    if (qb < 0) qb = 0;
    if (qe > (1ll << 29)) qe = 1ll << 29;
    if (qe <= qb) return 0;
    std::vector<uint32_t> want; reg2bins(qb, qe, want);
    std::vector<std::pair<uint64_t, uint64_t>> chunks; uint64_t min_off = 0;
    size_t o = 8;
    for (int t = 0; t <= tid; ++t) { // walk to the reference's section
        if (o + 4 > ix.size()) return io_err(-35, "truncated .bai");
        const int n_bin = le32(ix.data() + o); o += 4;
        for (int b = 0; b < n_bin; ++b) {
            if (o + 8 > ix.size()) return io_err(-35, "truncated .bai");
            const uint32_t bin = (uint32_t)le32(ix.data() + o); const int n_chunk = le32(ix.data() + o + 4); o += 8;
            if (o + 16ull * (size_t)n_chunk > ix.size()) return io_err(-35, "truncated .bai");
            if (t == tid && bin != 37450 && std::find(want.begin(), want.end(), bin) != want.end())
                for (int c = 0; c < n_chunk; ++c) chunks.emplace_back(le64(ix.data() + o + 16 * (size_t)c), le64(ix.data() + o + 16 * (size_t)c + 8));
            o += 16 * (size_t)n_chunk;
        }
        if (o + 4 > ix.size()) return io_err(-35, "truncated .bai");
        const int n_intv = le32(ix.data() + o); o += 4;
        if (o + 8ull * (size_t)n_intv > ix.size()) return io_err(-35, "truncated .bai");
        if (t == tid && n_intv > 0) min_off = le64(ix.data() + o + 8 * (size_t)std::min<int64_t>(qb >> 14, n_intv - 1));
        o += 8 * (size_t)n_intv;
    }
    std::vector<std::pair<uint64_t, uint64_t>> keep;
    for (auto &c : chunks) if (c.second > min_off) keep.push_back(c);
    std::sort(keep.begin(), keep.end());
    for (auto &c : keep) { if (!merged.empty() && c.first <= merged.back().second) merged.back().second = std::max(merged.back().second, c.second); else merged.push_back(c); }
    return 0;
}
// total length of the BGZF block at compressed offset c (0 at the end of the file, < 0 on a bad block)
static long long bgzf_block_len(FILE *f, uint64_t c) {
    uint8_t h[12];
    if (fseek(f, (long)c, SEEK_SET) != 0 || fread(h, 1, 12, f) != 12) return 0;
    if (h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) return io_err(-31, "not a BGZF block");
    const unsigned xlen = h[10] | (h[11] << 8);
    std::vector<uint8_t> x(xlen);
    if (xlen && fread(x.data(), 1, xlen, f) != xlen) return io_err(-31, "truncated BGZF header");
    for (size_t q = 0; q + 4 <= xlen;) { const unsigned slen = x[q + 2] | (x[q + 3] << 8); if (x[q] == 'B' && x[q + 1] == 'C' && slen == 2 && q + 6 <= xlen) return (long long)(x[q + 4] | (x[q + 5] << 8)) + 1; q += 4 + slen; }
    return io_err(-31, "BGZF block without BSIZE");
}
} // namespace

int lcd_io_region_image(const char *bam_path, const char *bai_path, const char *chrom, int64_t reg_beg, int64_t reg_end, LcdRegionImage &out) {
    BgzfStream bz; std::vector<std::pair<uint64_t, uint64_t>> merged;
    if (int rc = region_chunks(bam_path, bai_path, chrom, reg_beg, reg_end, bz, out.tid, out.tlen, out.n_ref, merged)) return rc;
    out.image.clear(); out.ranges.clear();
    if (merged.empty()) return 0;
    // whole blocks of the file: [first block of the chunk, the block its end lies in]
    struct Iv { uint64_t c0, c1, at; };
    std::vector<Iv> ivs;
    fseek(bz.f, 0, SEEK_END); const uint64_t fsize = (uint64_t)std::max<long>(ftell(bz.f), 0);
    for (auto &m : merged) {
        const uint64_t cb = std::min<uint64_t>(m.first >> 16, fsize), ce = std::min<uint64_t>(m.second >> 16, fsize); // (an index chunk may end behind the file's last block)
        uint64_t end = ce;
        if ((m.second & 0xffff) && ce < fsize) { const long long bl = bgzf_block_len(bz.f, ce); if (bl < 0) return (int)bl; end = ce + (uint64_t)bl; }
        if (!ivs.empty() && cb <= ivs.back().c1) ivs.back().c1 = std::max(ivs.back().c1, end);
        else ivs.push_back(Iv{cb, end, 0});
    }
    size_t total = 0;
    for (Iv &v : ivs) { v.at = total; total += (size_t)(v.c1 - v.c0); }
    out.image.resize(total);
    for (Iv &v : ivs) {
        if (v.c1 == v.c0) continue;
        if (fseek(bz.f, (long)v.c0, SEEK_SET) != 0 || fread(out.image.data() + v.at, 1, (size_t)(v.c1 - v.c0), bz.f) != (size_t)(v.c1 - v.c0)) return io_err(-31, "truncated BGZF block");
    }
    // block starts of the image -> offsets of its inflated stream
    std::vector<std::pair<uint64_t, uint64_t>> starts; // (offset in the image, inflated offset)
    {
        std::vector<Blk> blks; size_t u = 0;
        if (int rc = bgzf_blocks(Bytes{out.image.data(), out.image.size()}, blks, &u)) return rc;
        size_t o = 0, uu = 0; const std::vector<uint8_t> &f = out.image;
        while (o + 18 <= f.size()) { // (bgzf_blocks has checked every header)
            const unsigned xlen = f[o + 10] | (f[o + 11] << 8); unsigned bsize = 0;
            for (size_t x = o + 12; x + 4 <= o + 12 + xlen;) { const unsigned slen = f[x + 2] | (f[x + 3] << 8); if (f[x] == 'B' && f[x + 1] == 'C' && slen == 2) bsize = f[x + 4] | (f[x + 5] << 8); x += 4 + slen; }
            const size_t end = o + bsize + 1;
            starts.emplace_back(o, uu);
            uu += f[end - 4] | (f[end - 3] << 8) | (f[end - 2] << 16) | ((size_t)f[end - 1] << 24);
            o = end;
        }
        starts.emplace_back(f.size(), uu);
    }
    auto upos = [&](uint64_t v) -> uint64_t {
        const uint64_t c = std::min<uint64_t>(v >> 16, fsize);
        if ((v >> 16) >= fsize) v = c << 16;
        size_t k = 0; while (k + 1 < ivs.size() && c > ivs[k].c1) ++k; // (sorted, disjoint and not adjacent: the interval the block starts in, or ends)
        const uint64_t at = ivs[k].at + (c - ivs[k].c0);
        auto it = std::lower_bound(starts.begin(), starts.end(), std::make_pair(at, (uint64_t)0));
        const uint64_t u0 = it == starts.end() ? starts.back().second : it->second;
        const uint64_t u1 = (it == starts.end() || it + 1 == starts.end()) ? starts.back().second : (it + 1)->second;
        return std::min<uint64_t>(u0 + (v & 0xffff), std::max(u0, u1)); // (an offset at a block's end is the next block's start)
    };
    for (auto &m : merged) out.ranges.emplace_back(upos(m.first), upos(m.second));
    return 0;
}

extern "C" {

const char *lcd_io_last_error(void) { return g_io_err.c_str(); }

int lcd_bam_load_region(const char *bam_path, const char *chrom, int64_t reg_beg, int64_t reg_end, int min_mapq, int n_threads, lcd_bam_reads_t *out) {
    memset(out, 0, sizeof(*out));
    std::vector<uint8_t> file, d;
    if (int rc = read_file(bam_path, file)) return rc;
    if (int rc = bgzf_inflate_all(file, d, n_threads > 0 ? n_threads : (int)std::thread::hardware_concurrency())) return rc;
    if (d.size() < 12 || memcmp(d.data(), "BAM\1", 4) != 0) return io_err(-33, "not a BAM file");
    size_t o = 8 + (size_t)le32(d.data() + 4);
    const int n_ref = le32(d.data() + o); o += 4;
    int tid = -1; int64_t tlen = 0;
    std::vector<std::string> names; std::vector<int> lens;
    for (int i = 0; i < n_ref; ++i) {
        const int ln = le32(d.data() + o); o += 4;
        names.emplace_back((const char *)d.data() + o, (size_t)std::max(ln - 1, 0)); o += ln;
        lens.push_back(le32(d.data() + o)); o += 4;
        if (names.back() == chrom) { tid = i; tlen = lens.back(); }
    }
    if (tid < 0) return io_err(-34, std::string("contig not in the BAM header: ") + chrom);
    Collector col(tid, reg_beg, reg_end, min_mapq);
    while (o + 36 <= d.size()) {
        const int bs = le32(d.data() + o); const uint8_t *r = d.data() + o + 4;
        if (bs < 32 || o + 4 + (size_t)bs > d.size()) return io_err(-33, "truncated BAM record");
        o += 4 + (size_t)bs;
        { const int tk = col.take(r, (size_t)bs); if (tk == -2) return io_err(-33, "malformed BAM record (a field runs past the record, or a placeholder CIGAR without its CG tag)"); if (tk < 0) break; }
    }
    return col.finish(out, tid, tlen, n_ref);
}
void lcd_bam_reads_free(lcd_bam_reads_t *r) {
    free(r->pos0); free(r->end_pos); free(r->mapq); free(r->flag); free(r->n_cigar); free(r->qlen); free(r->cigar_off); free(r->cigar_pool);
    free(r->seq_off); free(r->seq_pool); free(r->qual_off); free(r->qual_pool); free(r->name_off); free(r->name_pool);
    memset(r, 0, sizeof(*r));
}

// The same region through the .bai (SAM specification 5.2): the chunks of the bins that can hold overlapping records, cut at the linear index's offset for
// the region's first 16 kb window, merged, and read in file order -- only those BGZF blocks are read and inflated.  Records, filters and order are those of
// lcd_bam_load_region (what sam_itr_queryi + the loop of collect_ref_seq_bam_main, src/bam_utils.c:1672-1706, see).
int lcd_bam_load_region_indexed(const char *bam_path, const char *bai_path, const char *chrom, int64_t reg_beg, int64_t reg_end, int min_mapq, lcd_bam_reads_t *out) {
    memset(out, 0, sizeof(*out));
    BgzfStream bz; int tid = -1, n_ref = 0; int64_t tlen = 0; uint8_t w[8];
    std::vector<std::pair<uint64_t, uint64_t>> merged;
    if (int rc = region_chunks(bam_path, bai_path, chrom, reg_beg, reg_end, bz, tid, tlen, n_ref, merged)) return rc;
    Collector col(tid, reg_beg, reg_end, min_mapq);
    std::vector<uint8_t> rec;
    bool done = false;
    for (size_t m = 0; m < merged.size() && !done; ++m) {
        if (int rc = bz.seek(merged[m].first)) return rc;
        for (;;) {
            if (int rc = bz.settle()) return rc;
            if (bz.eof || bz.tell() >= merged[m].second) break;
            const int r4 = bz.read(w, 4);
            if (r4 == 1) break;
            if (r4 < 0) return r4;
            const int bs = le32(w);
            if (bs < 32) return io_err(-33, "truncated BAM record");
            rec.resize((size_t)bs);
            if (bz.read(rec.data(), (size_t)bs) != 0) return io_err(-33, "truncated BAM record");
            { const int tk = col.take(rec.data(), (size_t)bs); if (tk == -2) return io_err(-33, "malformed BAM record (a field runs past the record, or a placeholder CIGAR without its CG tag)"); if (tk < 0) { done = true; break; } }
        }
    }
    return col.finish(out, tid, tlen, n_ref);
}

// faidx_fetch_seq([beg, end], 1-based inclusive) through the .fai next to the FASTA, as byte codes
int64_t lcd_fasta_fetch(const char *fa_path, const char *chrom, int64_t beg, int64_t end, uint8_t **codes_out) {
    *codes_out = nullptr;
    FILE *fi = fopen((std::string(fa_path) + ".fai").c_str(), "r");
    if (!fi) return io_err(-30, std::string("cannot open ") + fa_path + ".fai");
    char name[1024]; long long len = 0, off = 0, lb = 0, lw = 0; bool found = false;
    while (fscanf(fi, "%1023s %lld %lld %lld %lld", name, &len, &off, &lb, &lw) == 5) if (!strcmp(name, chrom)) { found = true; break; }
    fclose(fi);
    if (!found || lb <= 0 || lw < lb) return io_err(-34, std::string("contig not in the .fai: ") + chrom);
    if (beg < 1) beg = 1;
    if (end > len) end = len;
    if (end < beg) return 0;
    FILE *f = fopen(fa_path, "rb");
    if (!f) return io_err(-30, std::string("cannot open ") + fa_path);
    const int64_t n = end - beg + 1;
    uint8_t *out = (uint8_t *)malloc((size_t)n + 1);
    const long long first = off + (beg - 1) / lb * lw + (beg - 1) % lb, last = off + (end - 1) / lb * lw + (end - 1) % lb;
    std::vector<char> raw((size_t)(last - first + 1));
    fseek(f, (long)first, SEEK_SET);
    const size_t got = fread(raw.data(), 1, raw.size(), f);
    fclose(f);
    int64_t k = 0;
    for (size_t i = 0; i < got && k < n; ++i) if (raw[i] != '\n' && raw[i] != '\r') out[k++] = nt4c((unsigned char)raw[i]);
    if (k != n) { free(out); return io_err(-30, "FASTA shorter than its index says"); }
    *codes_out = out;
    return n;
}

// the lines write_vcf_header appends (src/vcf_utils.c:17-96), then the column line; *text_out malloc()'d
int lcd_vcf_header(const char *source_version, const char *cmdline, const char *date_yyyymmdd, int n_contigs, const char *const *contig_names, const int64_t *contig_lens,
                   const char *sample_name, char **text_out) {
    std::string t = "##fileformat=VCFv4.2\n";
    t += "##fileDate=" + std::string(date_yyyymmdd ? date_yyyymmdd : "") + "\n";
    t += "##source=longcallD version=" + std::string(source_version ? source_version : "") + "\n";
    t += "##CL=" + std::string(cmdline ? cmdline : "") + "\n";
    for (int i = 0; i < n_contigs; ++i) t += "##contig=<ID=" + std::string(contig_names[i]) + ",length=" + std::to_string((long long)contig_lens[i]) + ">\n";
    static const char *fixed[] = {
        "##FILTER=<ID=PASS,Description=\"All filters passed\">", "##FILTER=<ID=LowQual,Description=\"Low quality variant\">",
        "##FILTER=<ID=RefCall,Description=\"Reference call\">", "##FILTER=<ID=NoCall,Description=\"Site has depth=0 resulting in no call\">",
        "##INFO=<ID=END,Number=1,Type=Integer,Description=\"End position of the variant described in this record\">",
        "##INFO=<ID=SOMATIC,Number=0,Type=Flag,Description=\"Somatic/mosaic variant\">",
        "##INFO=<ID=CLEAN,Number=0,Type=Flag,Description=\"Clean-region variant (SNP or simple indel in non-repetitive region)\">",
        "##INFO=<ID=SVTYPE,Number=1,Type=String,Description=\"Type of structural variant\">",
        "##INFO=<ID=SVLEN,Number=A,Type=Integer,Description=\"Difference in length between REF and ALT alleles\">",
        "##INFO=<ID=TSD,Number=A,Type=String,Description=\"Target site duplication sequence\">",
        "##INFO=<ID=TSDLEN,Number=A,Type=Integer,Description=\"Length of target site duplication\">",
        "##INFO=<ID=POLYALEN,Number=A,Type=Integer,Description=\"Length of polyA/T sequence\">",
        "##INFO=<ID=MEI,Number=0,Type=Flag,Description=\"Mobile element insertion\">",
        "##INFO=<ID=TSDPOS1,Number=A,Type=Integer,Description=\"Start position of first target site duplication on CHROM\">",
        "##INFO=<ID=TSDPOS2,Number=A,Type=Integer,Description=\"Start position of second target site duplication on CHROM\">",
        "##INFO=<ID=REPNAME,Number=A,Type=String,Description=\"Repeat name\">",
        "##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">", "##FORMAT=<ID=GQ,Number=1,Type=Integer,Description=\"Genotype quality\">",
        "##FORMAT=<ID=DP,Number=1,Type=Integer,Description=\"Total read depth\">", "##FORMAT=<ID=AD,Number=R,Type=Integer,Description=\"Read depth for each allele\">",
        "##FORMAT=<ID=VAF,Number=A,Type=Float,Description=\"Variant allele frequency\">",
        "##FORMAT=<ID=PL,Number=G,Type=Integer,Description=\"Phred-scaled genotype likelihoods rounded to the closest integer\">",
        "##FORMAT=<ID=PS,Number=1,Type=Integer,Description=\"Phase set\">",
        "##FORMAT=<ID=ALTREADS,Number=.,Type=String,Description=\"IDs of reads supporting the variant\">"};
    for (const char *l : fixed) { t += l; t += '\n'; }
    t += "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + std::string(sample_name ? sample_name : "SAMPLE") + "\n";
    char *o = (char *)malloc(t.size() + 1);
    memcpy(o, t.c_str(), t.size() + 1);
    *text_out = o;
    return (int)std::count(t.begin(), t.end(), '\n');
}

// ---- BGZF blocks inflated on the device (inflate_kernel.hip): one upload of the compressed bytes, one wavefront per block, the inflated stream stays in HBM ----
struct InflateJobH { unsigned long long src, dst; unsigned clen, ulen, crc, pad_; };
struct InflateOutH { int status; unsigned crc; unsigned ulen; unsigned n_sym; unsigned long long t_total, t_tables, t_flush, t_match; };
} // extern "C"
void lcd_launch_inflate(const void *jobs, void *outs, int n_jobs, int verify, int timers, hipStream_t stream);
void lcd_inflate_set_x2n(const unsigned *t32, hipStream_t st);
namespace {
unsigned gf2_mulmod_h(unsigned a, unsigned b) { unsigned m = 1u << 31, p = 0; for (;;) { if (a & m) { p ^= b; if ((a & (m - 1)) == 0) break; } m >>= 1; b = (b & 1) ? (b >> 1) ^ 0xedb88320u : b >> 1; } return p; }
}
struct lcd_inflated_s { void *d_in = nullptr, *d_out = nullptr, *d_jobs = nullptr, *d_outs = nullptr; size_t total = 0, n_blocks = 0, comp_bytes = 0; double ms_kernel = 0, ms_h2d = 0;
                        long long accounted = 0; int device = 0; hipStream_t st = nullptr; hipEvent_t ev[3] = {nullptr, nullptr, nullptr}; };
extern "C" void lcd_account_device_bytes(int device, long long delta); // lcd_host.cpp: the library's device-memory ledger (LCD_MEM_FRACTION planning sees an inflated stream too)
extern "C" {
#define IOHIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { io_err(-40, std::string("HIP: ") + hipGetErrorString(e_) + " in " #x); lcd_inflated_free(h); return nullptr; } } while (0)
void lcd_inflated_free(lcd_inflated_t *h) {
    if (!h) return;
    if (h->d_in) (void)hipFree(h->d_in);
    if (h->d_out) (void)hipFree(h->d_out);
    if (h->d_jobs) (void)hipFree(h->d_jobs);
    if (h->d_outs) (void)hipFree(h->d_outs);
    for (auto &e : h->ev) if (e) (void)hipEventDestroy(e);
    if (h->st) (void)hipStreamDestroy(h->st);
    if (h->accounted) lcd_account_device_bytes(h->device, -h->accounted);
    delete h;
}
lcd_inflated_t *lcd_bgzf_inflate_dev(const uint8_t *file, size_t n, int verify_crc) {
    lcd_inflated_t *h = new lcd_inflated_s();
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { io_err(-40, "lcd_bgzf_inflate_dev: no HIP device (this entry point has no host path: lcd_bam_load_region inflates on host threads)"); delete h; return nullptr; }
    std::vector<Blk> blks; size_t total = 0;
    if (bgzf_blocks(Bytes{file, n}, blks, &total)) { delete h; return nullptr; }
    h->total = total; h->n_blocks = blks.size(); h->comp_bytes = n;
    if (blks.empty()) return h;
    (void)hipGetDevice(&h->device);
    IOHIP(hipMalloc(&h->d_in, n + 1024));            // (the decoder's 256-byte input windows read ahead of the stream: at most 512 bytes behind a block's end, inflate_kernel.hip BitIn::lim)
    IOHIP(hipMalloc(&h->d_out, total + 64));
    IOHIP(hipMalloc(&h->d_jobs, blks.size() * sizeof(InflateJobH)));
    IOHIP(hipMalloc(&h->d_outs, blks.size() * sizeof(InflateOutH)));
    h->accounted = (long long)(n + 1024 + total + 64 + blks.size() * (sizeof(InflateJobH) + sizeof(InflateOutH)));
    lcd_account_device_bytes(h->device, h->accounted);
    IOHIP(hipStreamCreateWithFlags(&h->st, hipStreamNonBlocking)); // (not the NULL stream: that one serialises with every blocking stream of the device)
    hipStream_t st = h->st; hipEvent_t (&ev)[3] = h->ev;
    for (auto &e : ev) IOHIP(hipEventCreate(&e));
    { // x^(2^k) mod P for the CRC combination: x^1 = 1 << 30 in the reflected representation, then squares
        unsigned t[32]; unsigned p = 1u << 30; t[0] = p;
        for (int k = 1; k < 32; ++k) t[k] = p = gf2_mulmod_h(p, p);
        lcd_inflate_set_x2n(t, st);
    }
    std::vector<InflateJobH> jobs(blks.size());
    for (size_t i = 0; i < blks.size(); ++i) {
        jobs[i].src = (unsigned long long)(uintptr_t)h->d_in + blks[i].cdata; jobs[i].dst = (unsigned long long)(uintptr_t)h->d_out + blks[i].uoff;
        jobs[i].clen = (unsigned)blks[i].clen; jobs[i].ulen = (unsigned)blks[i].ulen; jobs[i].crc = blks[i].crc; jobs[i].pad_ = 0;
    }
    IOHIP(hipEventRecord(ev[0], st));
    IOHIP(hipMemsetAsync((uint8_t *)h->d_in + n, 0, 1024, st));
    IOHIP(hipMemcpyAsync(h->d_in, file, n, hipMemcpyHostToDevice, st));
    IOHIP(hipMemcpyAsync(h->d_jobs, jobs.data(), jobs.size() * sizeof(InflateJobH), hipMemcpyHostToDevice, st));
    IOHIP(hipEventRecord(ev[1], st));
    const bool timers = getenv("LCD_INFLATE_PROFILE") != nullptr;
    lcd_launch_inflate(h->d_jobs, h->d_outs, (int)blks.size(), verify_crc, timers ? 1 : 0, st);
    IOHIP(hipGetLastError());
    IOHIP(hipEventRecord(ev[2], st));
    std::vector<InflateOutH> outs(blks.size());
    IOHIP(hipMemcpyAsync(outs.data(), h->d_outs, outs.size() * sizeof(InflateOutH), hipMemcpyDeviceToHost, st));
    IOHIP(hipStreamSynchronize(st));
    float a = 0, b = 0; (void)hipEventElapsedTime(&a, ev[0], ev[1]); (void)hipEventElapsedTime(&b, ev[1], ev[2]);
    h->ms_h2d = a; h->ms_kernel = b;
    for (auto &e : ev) { (void)hipEventDestroy(e); e = nullptr; }
    if (timers) {
        unsigned long long tt = 0, tb = 0, tf = 0, tm = 0, ns = 0;
        for (const InflateOutH &o : outs) { tt += o.t_total; tb += o.t_tables; tf += o.t_flush; tm += o.t_match; ns += o.n_sym; }
        fprintf(stderr, "[inflate] %zu blocks: %.0f symbols per block, ticks per block %.0f (tables %.1f %%, matches %.1f %%, flushes %.1f %%), %.1f ticks per symbol\n", outs.size(), (double)ns / outs.size(),
                (double)tt / outs.size(), 100.0 * tb / tt, 100.0 * tm / tt, 100.0 * tf / tt, (double)tt / (double)ns);
    }
    for (size_t i = 0; i < outs.size(); ++i) if (outs[i].status != 0) {
        static const char *why[] = {"ok", "?", "stored block: LEN / NLEN mismatch", "more output than ISIZE", "reserved block type", "too many codes", "bad code lengths", "no end-of-block code",
                                    "over-subscribed code", "invalid code", "invalid symbol", "distance before the start of the block", "fewer bytes than ISIZE", "CRC-32 mismatch",
                                    "the deflate stream runs past the block's compressed bytes (truncated or crafted block)"};
        io_err(-32, "device inflate failed on BGZF block " + std::to_string(i) + ": " + (outs[i].status < 15 ? why[outs[i].status] : "?"));
        lcd_inflated_free(h); return nullptr;
    }
    return h;
}
#undef IOHIP
uint64_t lcd_inflated_dev_ptr(const lcd_inflated_t *h) { return (uint64_t)(uintptr_t)h->d_out; }
size_t lcd_inflated_size(const lcd_inflated_t *h) { return h->total; }
size_t lcd_inflated_n_blocks(const lcd_inflated_t *h) { return h->n_blocks; }
double lcd_inflated_kernel_ms(const lcd_inflated_t *h) { return h->ms_kernel; }
double lcd_inflated_upload_ms(const lcd_inflated_t *h) { return h->ms_h2d; }
int lcd_inflated_to_host(const lcd_inflated_t *h, size_t off, size_t n, uint8_t *out) {
    if (off + n > h->total) return io_err(-41, "lcd_inflated_to_host: range past the end of the stream");
    if (n == 0) return 0;
    const hipError_t e = hipMemcpy(out, (const uint8_t *)h->d_out + off, n, hipMemcpyDeviceToHost);
    return e == hipSuccess ? 0 : io_err(-40, std::string("HIP: ") + hipGetErrorString(e));
}
}

