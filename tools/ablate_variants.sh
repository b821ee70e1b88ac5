set -u
export SE3_LIB_SUFFIX=_ab  # variant builds go to lib/libse3conv_hip_ab.so (se3conv3d_amd/build.py): the shipped library is never overwritten
SE3_CXXFLAGS="-DSE3_ABLATE_MASK=15" python -m se3conv3d_amd.build --force > /dev/null 2>&1
echo "mask=15 per-item FC2: $(SE3_NO_PAIR=1 timeout -k 10 200 python tools/profile_levels.py 2>&1 | grep -A1 'level 0' | tail -1)"
echo "mask=15 stream  FC2: $(SE3_NO_PAIR=1 SE3_STREAM=1 timeout -k 10 200 python tools/profile_levels.py 2>&1 | grep -A1 'level 0' | tail -1)"
echo "mask=15 pair       : $(timeout -k 10 200 python tools/profile_levels.py 2>&1 | grep -A1 'level 0' | tail -1)"
