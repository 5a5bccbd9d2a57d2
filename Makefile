# Top-level convenience targets.  `make` builds what __graft_entry__.build() builds.
PY ?= python3
SAN = -O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined
SANLIBS = $(shell gcc -print-file-name=libasan.so):$(shell gcc -print-file-name=libubsan.so)

all:
	$(PY) -c "import __graft_entry__ as g; g.build()"

# AddressSanitizer + UBSan over the HOST code that shares its arithmetic with the kernels (csrc/host_check.cpp: the field,
# curve and host-finishing headers compiled with g++) and over the C oracle; runs the CPU tests that drive both.  CPU only:
# no device code is instrumented and no GPU is touched.  The normal builds are restored afterwards, so no instrumented
# library is left in the tree.
asan:
	$(MAKE) -B -C davinci-node_amd/csrc libdvn_hostcheck.so HOSTCHECK_FLAGS="$(SAN)"
	$(MAKE) -B -C oracle liboracle_bn254.so CFLAGS="$(SAN) -fopenmp -fPIC -Wall -Wno-unused-function"
	LD_PRELOAD=$(SANLIBS) ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
		$(PY) -m pytest tests/test_host_arith.py tests/test_oracle.py tests/test_groth16.py -m "not gpu" -q -x -p no:cacheprovider; \
		rc=$$?; \
		$(MAKE) -B -C davinci-node_amd/csrc libdvn_hostcheck.so; \
		$(MAKE) -B -C oracle liboracle_bn254.so; \
		exit $$rc

.PHONY: all asan
