# two processes tracing 1-spp iterations on the same GPU at once: what overlapping consecutive launches could give
one() { python3 bench.py --config c2 --batch 1 --steps 3000 --warmup 50 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(d["value"], d["ms_per_step"])'; }
echo "alone: $(one)"
one > /tmp/a.txt & one > /tmp/b.txt & wait
echo "two at once: $(cat /tmp/a.txt) | $(cat /tmp/b.txt)"
one > /tmp/a.txt & one > /tmp/b.txt & one > /tmp/c.txt & wait
echo "three at once: $(cat /tmp/a.txt) | $(cat /tmp/b.txt) | $(cat /tmp/c.txt)"
