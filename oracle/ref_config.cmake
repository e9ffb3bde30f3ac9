# Produces oracle/_ref/config.h from the reference's own template (config.h.in) with
# CMake's configure_file() in script mode.  The option values are the reference's
# defaults (reference CMakeLists.txt:9-24) except BUILD_VISUALIZATION=OFF (GLUT is not
# installed in the image; visualisation is outside the hot path).  The reference's own
# CMakeLists.txt is NOT executed; every object is compiled by gcc from oracle/Makefile.
set(BUILD_AVX2 ON)
set(FIVE_FIXRES ON)
set(FIVE_NONAN ON)
set(FIVE_NOINF ON)
set(FRIRL_FAST ON)
set(DOUBLE_PRECISION ON)
set(FAST_ABS ON)
set(FAST_POW ON)
set(FAST_SQRT ON)
set(BUILD_CHECK_STATES ON)
set(DEBUG OFF)
set(PREDICT_BRANCHES OFF)
set(BUILD_OPENMP OFF)
set(BUILD_MPI OFF)
set(BUILD_VISUALIZATION OFF)
configure_file(${REF}/config.h.in ${OUT}/config.h)
