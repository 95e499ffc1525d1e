"""full-size configs[1] batch on the GPU vs the oracle, region by region (debugging aid)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from multiprocessing import get_context
from longcalld_amd import jobs

SEED, N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000, int(sys.argv[2]) if len(sys.argv) > 2 else 1250
SHAPE = jobs.ONT if len(sys.argv) > 3 and sys.argv[3] == "ont" else jobs.HIFI


def orc_region(k):
    from oracle import pyoracle as orc
    regs = jobs.make_regions(SEED, N, SHAPE)
    return k, orc.collect_noisy_reg_aln_strs(regs[k])


def main():
    from oracle import pyoracle as orc
    orc.build()
    from longcalld_amd import align as lcd
    from conftest import same_result
    regs = jobs.make_regions(SEED, N, SHAPE)
    o = lcd.default_opt(); o.is_ont = 1 if SHAPE is jobs.ONT else 0; o.collect_noisy_vars = 1; o.collect_ref_read_aln_str = 1
    b = lcd.RegionBatch(o)
    for r in regs:
        b.add_region(r)
    b.upload(); b.run(); b.download()
    got = [b.result(k) for k in range(N)]
    st = b.stats()
    print("retries", st["poa_retries"], "resolved", st["n_regions_resolved"])
    bad = []
    for k in range(N):
        oo = orc.default_opt(); oo.collect_ref_read_aln_str = 1
        exp = orc.collect_noisy_reg_aln_strs(regs[k], oo)
        try:
            same_result(exp, got[k])
        except AssertionError as e:
            bad.append(k)
            if len(bad) <= 5:
                print("MISMATCH region", k, "len", regs[k]["reg_len"], "reads", len(regs[k]["seqs"]), "n_cons", exp["n_cons"], got[k]["n_cons"], str(e)[:200])
    print("mismatching regions:", len(bad), bad[:20])


if __name__ == "__main__":
    main()
