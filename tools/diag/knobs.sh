# per-kernel HIP-event time of a 2^20 proof under a list of BPG_* settings (tools/diag/kprof.py); usage: knobs.sh "VAR=val VAR2=val" ...
export BPG_PROFILE=serving
for kv in "$@"; do
  echo "== $kv"
  env $kv timeout -k 10 120 python3 tools/diag/kprof.py 512 3 1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernels']
grp={'sweep':['k_bucket_chunks'],'epilogue':['k_bucket_combine','k_bucket_combine_heavy','k_bucket_reduce','k_window_sums'],'sort':['k_msm_digits','k_msm_count1','k_msm_scatter1','k_msm_sort2','k_scan_blocksums','k_scan_top','k_scan_apply'],'folds':['k_fold_points_wnaf','k_fold_points_quad','k_fold_points_reg','k_fold_points_split','k_fold_points','k_normalize_niels'],'tail':['k_tt_round','k_tt_round8','k_tt_finish','k_tt_bases','k_tt_multiples','k_tt_advance','k_tt_factors']}
tot={g:sum(k.get(n,[0,0])[1] for n in ns) for g,ns in grp.items()}
rest=d['gpu_ms']-sum(tot.values())
print('gpu_ms %.2f wall %.1f | '%(d['gpu_ms'],d['wall_ms'])+' '.join('%s %.2f'%(g,v) for g,v in tot.items())+' rest %.2f'%rest)
print('   '+' '.join('%s %d x %.3f'%(n,v[0],v[1]) for n,v in list(k.items())[:16]))
"
done
