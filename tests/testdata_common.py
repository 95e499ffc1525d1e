"""Reader of tests/golden/testdata_chunk.npz (made by tests/golden/make_testdata_fixture.py from the reference's bundled
test_data): one real HG002 HiFi chunk -- reads as digar lists + 4-bit bases + qualities, the reference slice, noisy regions with their
read lists, a read x variant profile for K5, and the oracle's expected outputs."""
import hashlib
import os

import numpy as np

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "testdata_chunk.npz")
NT16 = np.array([4, 0, 1, 4, 2, 4, 4, 4, 3, 4, 4, 4, 4, 4, 4, 4], np.uint8)  # htslib seq_nt16_int


class Chunk:
    def __init__(self, arrays=None):
        z = arrays if arrays is not None else np.load(PATH)
        self.z = {k: z[k] for k in (z.keys() if isinstance(z, dict) else z.files)}
        z = self.z
        self.n_reads = len(z["read_qlen"])
        self.regions = [(int(b), int(e)) for b, e in z["regions"]]
        self.digars = [z["digars"][z["digar_off"][i]:z["digar_off"][i + 1]] for i in range(self.n_reads)]
        self.bseq = [z["bseq"][z["bseq_off"][i]:z["bseq_off"][i + 1]] for i in range(self.n_reads)]
        self.qual = [z["qual"][z["qual_off"][i]:z["qual_off"][i + 1]] for i in range(self.n_reads)]
        self.qlen = z["read_qlen"]

    def reg_reads(self, k):
        return self.z["reg_reads"][self.z["reg_read_off"][k]:self.z["reg_read_off"][k + 1]]

    def ref_slice(self, k):
        b, e = self.regions[k]
        o = int(self.z["ref_beg"])
        return self.z["ref"][b - o:e - o + 1]

    def bases(self, i, rb, re):
        j = np.arange(rb, re + 1)
        return NT16[(self.bseq[i][j >> 1] >> ((~j & 1) << 2)) & 0xf]

    def hap_problem(self):
        p = {k[4:]: v for k, v in self.z.items() if k.startswith("hap_")}
        for k in ("n_reads", "n_vars", "is_ont"):
            p[k] = int(p[k])
        return p

    def region_dict(self, orc, k, haps, pss, flank=10):
        """the region after collect_noisy_read_info, sliced with the ORACLE's digar walk (input of the oracle's region driver)"""
        b, e = self.regions[k]
        ids = self.reg_reads(k)
        seqs, quals, covers = [], [], []
        for i in ids:
            rb, re, cv = orc.read_region_slice(self.digars[i], self.qlen[i], b, e, flank)
            seqs.append(self.bases(i, rb, re) if re >= rb else np.zeros(0, np.uint8))
            quals.append(self.qual[i][rb:re + 1].copy() if re >= rb else np.zeros(0, np.uint8))
            covers.append(cv)
        return dict(reg_len=e - b + 1, read_ids=ids.astype(np.int32), seqs=seqs, quals=quals, covers=np.array(covers, np.int32),
                    haps=np.asarray(haps, np.int32)[ids], phase_sets=np.asarray(pss, np.int64)[ids], ref=self.ref_slice(k))


def result_digest(res):
    """64-bit digest of one region result dict (n_cons, clusters, every alignment row and coordinate)"""
    h = hashlib.blake2b(digest_size=8)
    h.update(np.int32(res["n_cons"]).tobytes())
    for c in range(res["n_cons"]):
        h.update(np.int32(res["clu_n_seqs"][c]).tobytes())
        h.update(np.ascontiguousarray(res["clu_read_ids"][c], np.int32).tobytes())
        for s in res["aln_strs"][c]:
            if s is None:
                h.update(b"\xff")
                continue
            h.update(np.array([s["aln_len"], s["target_beg"], s["target_end"], s["query_beg"], s["query_end"]], np.int32).tobytes())
            h.update(np.ascontiguousarray(s["target"], np.uint8).tobytes()); h.update(np.ascontiguousarray(s["query"], np.uint8).tobytes())
    return int.from_bytes(h.digest(), "little")
