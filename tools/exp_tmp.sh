for L in 2 3; do for K in 20 200; do
python bench.py --steps $K --warmup 5 --lanes $L --no-cpu-baseline --no-sequential-leg --no-extra-legs --no-fp32-leg 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('mixed lanes $L K $K:', d['value'], d['first_item_latency_ms'])"
done; done
VIDC_PRECISION=mixed python tools/group_timeline.py 20 2 2 | grep -E "===|lane [0-9]:"
VIDC_PRECISION=mixed python tools/group_timeline.py 20 3 2 | grep -E "===|lane [0-9]:"
