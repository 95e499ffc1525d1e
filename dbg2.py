import sys; sys.path.insert(0,'.')
import numpy as np, os
from longcalld_amd import align as A, jobs
rng=np.random.default_rng(3)
reg=jobs.make_region(rng, jobs.HIFI, length=3000, n_reads=38)
reg['haps'][:]=np.arange(38)%2+1; reg['phase_sets'][:]=777; reg['covers'][:]=12
reg['seqs']=[s if c==12 else s for s,c in zip(reg['seqs'],reg['covers'])]
b=A.RegionBatch(); b.add_region(reg); b.upload(); b.run(); b.run()
print(b.stats()['ms_poa_kernel'], b.stats()['poa_cells'])
