# Convenience targets; `python -c "import __graft_entry__ as g; g.build()"` does the same from Python.
HIPCC ?= /opt/rocm/bin/hipcc
PKG   := project3-cuda-path-tracer_amd
HIPFLAGS := --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -std=c++17

all: lib host oracle

lib: $(PKG)/libptmi355.so
$(PKG)/libptmi355.so: $(PKG)/csrc/ptmi355.hip $(wildcard $(PKG)/csrc/*.hpp) include/ptmi355.h
	$(HIPCC) $(HIPFLAGS) -o $@ $<

host: lib
	python3 -c "import __graft_entry__ as g; p = g.load_package(); p.build_host(); p.build_ptbench()"

oracle:
	$(MAKE) -C oracle all

test:
	python3 -m pytest tests -q -m "not gpu"

test-gpu:
	python3 -m pytest tests -q -m gpu

bench:
	python3 bench.py

clean:
	rm -f $(PKG)/*.so $(PKG)/ptbench
	$(MAKE) -C oracle clean

.PHONY: all lib host oracle test test-gpu bench clean
