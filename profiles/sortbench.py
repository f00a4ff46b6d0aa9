import sys; sys.path.insert(0,'.')
import __graft_entry__ as e
K=e.load_package(); c=K.Context()
for n in (238_000_000,):
    ms,msl,inv=c.selftest_sort(n,3)
    print("n=%d sort %.3f ms, scatter launch %.3f ms -> %.1f GB/s per launch, inversions %d"%(n,ms,msl,n*32/msl/1e6,inv))
