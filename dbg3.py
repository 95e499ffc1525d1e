import sys; sys.path.insert(0,'.')
import numpy as np, os
from longcalld_amd import align as A, jobs
rng=np.random.default_rng(3)
reg=jobs.make_region(rng, jobs.HIFI, length=3600, n_reads=30)
reg['haps'][:]=0; reg['phase_sets'][:]=-1; reg['covers'][:]=12
b=A.RegionBatch(); b.add_region(reg); b.upload()
try:
    b.run(); b.run()
    print(b.stats()['ms_poa_kernel'], b.stats()['poa_cells'])
except Exception as e: print('ERR', e)
