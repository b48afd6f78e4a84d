# oracle/_ref: the part of the reference that builds in this image from its own sources.
#
# Only src/vacancy/marching_cubes_lut.cc qualifies: every other file of the path includes
# Eigen/Geometry (include/vacancy/common.h:21) from an empty, un-vendored submodule, Eigen is not
# installed, and writing a stand-in header is not allowed -- those files are unbuildable here.
# The LUT file is compiled UNMODIFIED from where it lies (nothing is copied into this repo); outputs
# go to oracle/_ref/ only (git-ignored, travels to the GPU box with the snapshot).
#   make -f ref.mk            (from oracle/)
REF ?= /root/reference
CXX ?= g++
OUT = _ref
CXXFLAGS = -std=c++14 -O2 -fPIC -I$(REF)/src -I$(REF)/include

all: $(OUT)/ref_lut_dump $(OUT)/libref_mc_lut.so

$(OUT)/marching_cubes_lut.o: $(REF)/src/vacancy/marching_cubes_lut.cc
	mkdir -p $(OUT)
	$(CXX) $(CXXFLAGS) -c $< -o $@

$(OUT)/ref_lut_dump: ref_lut_dump.cc $(OUT)/marching_cubes_lut.o
	$(CXX) $(CXXFLAGS) -DREF_LUT_DUMP_MAIN -o $@ ref_lut_dump.cc $(OUT)/marching_cubes_lut.o

$(OUT)/libref_mc_lut.so: ref_lut_dump.cc $(OUT)/marching_cubes_lut.o
	$(CXX) $(CXXFLAGS) -shared -o $@ ref_lut_dump.cc $(OUT)/marching_cubes_lut.o

clean:
	rm -rf $(OUT)
