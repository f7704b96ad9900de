# iris_lamaConfig.cmake -- lets a consumer of the reference (iris_lama_ros: `find_package(iris_lama REQUIRED)`,
# `target_link_libraries(node iris_lama::iris_lama)`, reference CMakeLists.txt:25-55 and
# cmake/iris_lamaConfig.cmake.in:7-11) pick up the MI355X path instead, in-tree:
#     cmake -Diris_lama_DIR=<this repo>/cmake ...
# The imported target is the host library liblama_host.so (it dlopen()s its sibling liblama_hip.so -- liblama_hip_wide.so for an l2_max beyond 127 cells -- at run time);
# build both first with `make -C iris_lama_amd` (or python -c "import __graft_entry__ as g; g.build()").
# Eigen3 is optional here: when it is found the public vector types are Eigen's (include/lama/types.h), otherwise the
# POD stand-ins are used.
get_filename_component(_lama_root "${CMAKE_CURRENT_LIST_DIR}/.." ABSOLUTE)
set(iris_lama_INCLUDE_DIRS "${_lama_root}/include")
set(_lama_host "${_lama_root}/iris_lama_amd/lib/liblama_host.so")
if(NOT EXISTS "${_lama_host}")
  set(iris_lama_FOUND FALSE)
  set(iris_lama_NOT_FOUND_MESSAGE "liblama_host.so is not built: run `make -C ${_lama_root}/iris_lama_amd`")
  return()
endif()
if(NOT TARGET iris_lama::iris_lama)
  add_library(iris_lama::iris_lama SHARED IMPORTED)
  set_target_properties(iris_lama::iris_lama PROPERTIES
    IMPORTED_LOCATION "${_lama_host}"
    INTERFACE_INCLUDE_DIRECTORIES "${iris_lama_INCLUDE_DIRS}"
    INTERFACE_COMPILE_FEATURES cxx_std_14)
  find_package(Eigen3 3.3 QUIET NO_MODULE)
  if(TARGET Eigen3::Eigen)
    set_property(TARGET iris_lama::iris_lama APPEND PROPERTY INTERFACE_LINK_LIBRARIES Eigen3::Eigen)
    set_property(TARGET iris_lama::iris_lama APPEND PROPERTY INTERFACE_COMPILE_DEFINITIONS LAMA_USE_EIGEN)
  endif()
  find_package(Threads QUIET)
  if(TARGET Threads::Threads)
    set_property(TARGET iris_lama::iris_lama APPEND PROPERTY INTERFACE_LINK_LIBRARIES Threads::Threads)
  endif()
  set_property(TARGET iris_lama::iris_lama APPEND PROPERTY INTERFACE_LINK_LIBRARIES ${CMAKE_DL_LIBS})
endif()
set(iris_lama_LIBRARIES iris_lama::iris_lama)
set(iris_lama_FOUND TRUE)
