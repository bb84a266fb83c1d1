.PHONY: build metaseg meta_overlay test clean

# same targets and config.yaml surface as the reference (Makefile:6-10); `build` compiles the gfx950 library first
build:
	python -m ecseg_amd.build

metaseg: build
	python src/metaseg.py

meta_overlay: build
	python src/meta_overlay.py

test:
	python -m pytest tests -q -m "not gpu"

clean:
	rm -rf __pycache__ ecseg_amd/csrc/*.o ecseg_amd/libecseg_hip.so
