# round 4: the rejected case solved ahead -- parity first, then the bench worker (quick form) for the A/B keys
timeout 600 python -m pytest tests/test_backend_gpu.py -q -p no:cacheprovider -k "solved_ahead or reuse_after or kept_state or optimize" 2>&1 | tail -5
timeout 600 python bench.py --worker --no-cpu --quick --steps 200 --warmup 20 > gpurun_out/ab_spec.json 2> gpurun_out/ab_spec.err
tail -3 gpurun_out/ab_spec.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/ab_spec.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "body", d.get("iteration_us"), "accepted", d.get("accepted_fraction"))
for k in ("value_with_literal_relinearize_on_reject", "value_with_system_reuse_after_rejected_steps", "value_without_rejected_case_solved_ahead"):
    print(k, d.get(k))
print(d.get("other_windows_same_protocol")); print(d.get("one_window_soak"))
PY
