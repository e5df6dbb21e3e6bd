for rep in 1 2; do for G in 0 1; do for L in 2 3; do
VIDC_GROUPED_SCHEDULER=$G python bench.py --steps 20 --warmup 5 --lanes $L --frames-per-launch 1 --no-cpu-baseline --no-sequential-leg --no-extra-legs 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('F=1 grouped=$G lanes $L K20: fp32', d['value'], ' mixed', d['value_mixed'])"
done; done; done
for G in 0 1; do
VIDC_GROUPED_SCHEDULER=$G python bench.py --batch 8 --source 640x480 --height 240 --plane-head --steps 40 --warmup 6 --frames-per-launch 1 --lanes 2 --no-cpu-baseline --no-sequential-leg --no-extra-legs 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('configs[2] grouped=$G: fp32', d['value'], ' mixed', d['value_mixed'])"
VIDC_GROUPED_SCHEDULER=$G python bench.py --batch 4 --source 1280x720 --height 240 --steps 60 --warmup 6 --frames-per-launch 1 --lanes 2 --no-cpu-baseline --no-sequential-leg --no-extra-legs 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('configs[3] grouped=$G: fp32', d['value'], ' mixed', d['value_mixed'])"
done
