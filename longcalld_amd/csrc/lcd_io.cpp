// lcd_io.cpp -- SURVEY 8(f) f3: the data formats in front of the hot path, without htslib (absent from the reference checkout).
//   * BGZF container + BAM records: what collect_ref_seq_bam_main (src/bam_utils.c:1659-1716) loads for one region -- primary, mapped reads of
//     MAPQ >= min_mq overlapping [reg_beg, reg_end], in file order -- flattened to exactly the arrays lcd_digar_batch and lcd_read_view_t take
//     (0-based position, CIGAR words, BAM 4-bit bases, qualities).  BGZF blocks are independent deflate streams: block boundaries come from the
//     BSIZE fields, the blocks are inflated on host threads in parallel (zlib), records are decoded from the concatenation.  No .bai: a
//     region is a scan (the index only saves I/O; results are those of sam_itr_queryi on (reg_beg - 1, reg_end]).
//   * FASTA + .fai: faidx_fetch_seq of a region as byte codes (nst_nt4_table).
//   * the VCF header lines of write_vcf_header (src/vcf_utils.c:17-96).  htslib's bcf_hdr_write decides their final order and adds
//     ##fileformat itself; that library is absent, so the header text is this project's rendering (the BODY lines are lcd_format_vcf's, exact).
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
struct Blk { size_t cdata, clen, uoff, ulen; };
// BGZF: gzip members with an extra field 'B','C',2,BSIZE (total block size - 1); ISIZE (last 4 bytes) = uncompressed length
int bgzf_blocks(const std::vector<uint8_t> &f, std::vector<Blk> &blks, size_t *total) {
    size_t o = 0, u = 0;
    while (o + 18 <= f.size()) {
        if (f[o] != 31 || f[o + 1] != 139 || f[o + 2] != 8 || !(f[o + 3] & 4)) return io_err(-31, "not a BGZF block");
        const unsigned xlen = f[o + 10] | (f[o + 11] << 8);
        size_t x = o + 12; unsigned bsize = 0; bool found = false;
        while (x + 4 <= o + 12 + xlen) {
            const unsigned slen = f[x + 2] | (f[x + 3] << 8);
            if (f[x] == 'B' && f[x + 1] == 'C' && slen == 2) { bsize = f[x + 4] | (f[x + 5] << 8); found = true; }
            x += 4 + slen;
        }
        if (!found || o + bsize + 1 > f.size()) return io_err(-31, "BGZF block without BSIZE / truncated file");
        const size_t end = o + bsize + 1;
        const unsigned isize = f[end - 4] | (f[end - 3] << 8) | (f[end - 2] << 16) | ((unsigned)f[end - 1] << 24);
        Blk b; b.cdata = o + 12 + xlen; b.clen = end - 8 - b.cdata; b.uoff = u; b.ulen = isize;
        if (isize) blks.push_back(b);
        u += isize; o = end;
    }
    *total = u;
    return 0;
}
int bgzf_inflate_all(const std::vector<uint8_t> &f, std::vector<uint8_t> &out, int n_threads) {
    std::vector<Blk> blks; size_t total = 0;
    if (int rc = bgzf_blocks(f, blks, &total)) return rc;
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
} // namespace

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
    std::vector<int64_t> pos0, endp; std::vector<int> mapq, flag, ncig, qlen;
    std::vector<uint64_t> coff, soff, qoff, noff; std::vector<uint32_t> cpool; std::vector<uint8_t> spool, qpool; std::vector<char> npool;
    while (o + 36 <= d.size()) {
        const int bs = le32(d.data() + o); const uint8_t *r = d.data() + o + 4;
        if (bs < 32 || o + 4 + (size_t)bs > d.size()) return io_err(-33, "truncated BAM record");
        o += 4 + (size_t)bs;
        const int refid = le32(r), p = le32(r + 4), lname = r[8], mq = r[9], nc = r[12] | (r[13] << 8), fl = r[14] | (r[15] << 8), lseq = le32(r + 16);
        if (refid != tid) { if (refid > tid && !pos0.empty()) break; continue; }
        const uint8_t *cg = r + 32 + lname;
        int64_t rl = 0;
        for (int k = 0; k < nc; ++k) { const uint32_t c = (uint32_t)le32(cg + 4 * k); const int op = c & 0xf; if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rl += c >> 4; }
        const int64_t e0 = p + (rl > 0 ? rl : 1); // bam_endpos: 0-based exclusive end (an alignment without reference bases spans one)
        if (p >= reg_end) break;                  // sorted input: nothing further overlaps (reg_beg - 1, reg_end]
        if (e0 <= reg_beg - 1) continue;
        if ((fl & (0x4 | 0x100 | 0x800)) || mq < min_mapq) continue; // BAM_FUNMAP | BAM_FSECONDARY | BAM_FSUPPLEMENTARY, src/bam_utils.c:1683
        pos0.push_back(p); endp.push_back(e0); mapq.push_back(mq); flag.push_back(fl); ncig.push_back(nc); qlen.push_back(lseq);
        coff.push_back(cpool.size()); for (int k = 0; k < nc; ++k) cpool.push_back((uint32_t)le32(cg + 4 * k));
        const uint8_t *sq = cg + 4 * (size_t)nc, *ql = sq + (lseq + 1) / 2;
        soff.push_back(spool.size()); spool.insert(spool.end(), sq, sq + (lseq + 1) / 2);
        qoff.push_back(qpool.size()); qpool.insert(qpool.end(), ql, ql + lseq);
        noff.push_back(npool.size()); npool.insert(npool.end(), (const char *)r + 32, (const char *)r + 32 + lname);
    }
    out->n_reads = (int)pos0.size(); out->tid = tid; out->target_len = tlen; out->n_targets = n_ref;
    out->pos0 = dup_vec(pos0); out->end_pos = dup_vec(endp); out->mapq = dup_vec(mapq); out->flag = dup_vec(flag); out->n_cigar = dup_vec(ncig); out->qlen = dup_vec(qlen);
    out->cigar_off = dup_vec(coff); out->cigar_pool = dup_vec(cpool); out->seq_off = dup_vec(soff); out->seq_pool = dup_vec(spool);
    out->qual_off = dup_vec(qoff); out->qual_pool = dup_vec(qpool); out->name_off = dup_vec(noff); out->name_pool = dup_vec(npool);
    return out->n_reads;
}
void lcd_bam_reads_free(lcd_bam_reads_t *r) {
    free(r->pos0); free(r->end_pos); free(r->mapq); free(r->flag); free(r->n_cigar); free(r->qlen); free(r->cigar_off); free(r->cigar_pool);
    free(r->seq_off); free(r->seq_pool); free(r->qual_off); free(r->qual_pool); free(r->name_off); free(r->name_pool);
    memset(r, 0, sizeof(*r));
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
}
