#!/bin/bash
# (experiment) how the incremental re-sort fares on a shape: exp/lib_vinc.so (-DLCD_X_VERIFY_INC -DLCD_X_INCSTAT).  Usage: bash tools/gpu_incstat.sh <tag> <bench flags...>
tag=$1; shift
mkdir -p gpurun_out
cp longcalld_amd/liblcd_hotpath.so /tmp/lib_orig.so
cp exp/lib_vinc.so longcalld_amd/liblcd_hotpath.so
timeout 900 python bench.py "$@" --warmup 1 --cpu-sample 0 --repeats 1 --depth-profile 0 --f3 0 --overlap 0 > gpurun_out/${tag}.out 2> gpurun_out/${tag}.err
cp /tmp/lib_orig.so longcalld_amd/liblcd_hotpath.so
grep -c "verify-inc" gpurun_out/${tag}.out
for m in 0 1; do
grep "\[inc\]" gpurun_out/${tag}.out | awk -v m=$m '$5==m {nc++; rd+=$7; nd+=$9; for(i=1;i<=NF;i++){if($i=="pre"){pre+=$(i+1)} if($i=="ml"){ml+=$(i+1)} if($i=="new"){nw+=$(i+1)} if($i=="cnt"){cnt+=$(i+1)} if($i=="cut"){cut+=$(i+1)} if($i=="cap"){cap+=$(i+1)} if($i=="q"){q+=$(i+1)} if($i=="dry"){dry+=$(i+1)} if($i=="el"){el+=$(i+1)} if($i=="left"){left+=$(i+1)} if($i=="ok"){ok+=$(i+1)} if($i=="walked"){wk+=$(i+1)} if($i=="pieces"){pc+=$(i+1)}}} END {print "mode",m,":",nc,"chains",rd,"reads, nodes/chain",nd/(nc+1e-9),": pre",pre,"ml",ml,"new",nw,"cnt",cnt,"cut",cut,"cap",cap,"q",q,"dry",dry,"el",el,"left",left,"| ok",ok,"walked",wk,"pieces",pc}'
done
python -c "import json; j=json.loads([l for l in open('gpurun_out/${tag}.out') if l.startswith('{')][-1]); print('verify build', j['value'], j['digest'])"
