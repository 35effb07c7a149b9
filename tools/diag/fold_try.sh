for kv in "BPG_NOOP=1" "BPG_FOLD_WNAF=4 BPG_FOLD_PARTS=4" "BPG_FOLD_WNAF=5 BPG_FOLD_PARTS=2" "BPG_FOLD_WNAF=4 BPG_FOLD_PARTS=2"; do
  echo "== $kv"; env $kv timeout -k 10 200 python3 tools/diag/lone_proof.py oneshot spin 3 2>/dev/null | python3 -c "
import json,sys; d=json.load(sys.stdin); v=list(d.values())[0]; print(v['phase_ms']['ipa'], v['phase_ms']['ipa_fold'], v['kernel_ms_sum'], v['top'].get('k_fold_points_wnaf'))"
done
