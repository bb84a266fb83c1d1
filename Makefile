.PHONY: build metaseg meta_overlay test asan clean

# same targets and config.yaml surface as the reference (Makefile:6-10); `build` compiles the gfx950 library first
build:
	python -m ecseg_amd.build

metaseg: build
	python src/metaseg.py

meta_overlay: build
	python src/meta_overlay.py

test:
	python -m pytest tests -q -m "not gpu"

# AddressSanitizer + UBSan build of the host-side codecs (the only code that parses untrusted bytes) with a corrupt-stream
# corpus; CPU only (GPU sanitizers are not available on the target pool)
asan:
	g++ -std=c++17 -O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined -fno-sanitize-recover=all \
	    ecseg_amd/csrc/host_codec.cpp ecseg_amd/csrc/host_io.cpp tools/asan/codec_fuzz.cpp -lz -o /tmp/ecseg_codec_fuzz
	/tmp/ecseg_codec_fuzz

clean:
	rm -rf __pycache__ ecseg_amd/csrc/*.o ecseg_amd/libecseg_*.so
