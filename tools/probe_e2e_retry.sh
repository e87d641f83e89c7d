python - <<PY
import torch
x = [torch.empty(150 * (1 << 28) // 8, dtype=torch.int32, device="cuda").fill_(1) for _ in range(8)]
torch.cuda.synchronize()
PY
python bench.py --no-cpu --no-copy --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); e=d['end_to_end']; print(e['ms_per_call'], e['attempts_ms'], e['value'])"
python bench.py --no-cpu --no-copy --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); e=d['end_to_end']; print(e['ms_per_call'], e['attempts_ms'], e['value'])"
