# Reference-side addon: binds grl's plug-in API to libgrlx.so (include/grlx.h of the grl_amd repository).
# Drop this directory into <grl>/addons/grlx; point GRLX_ROOT at the grl_amd checkout (include/ + grl_amd/lib/).
set(TARGET addon_grlx)

find_library(GRLX_LIBRARY grlx HINTS $ENV{GRLX_ROOT}/grl_amd/lib)
find_path(GRLX_INCLUDE_DIR grlx.h HINTS $ENV{GRLX_ROOT}/include)

if (GRLX_LIBRARY AND GRLX_INCLUDE_DIR)
  set(GRL_BUILD_GRLX ON CACHE BOOL "Build MI355X (libgrlx) addon")
else()
  message("** Cannot build grlx addon: libgrlx.so / grlx.h not found (set GRLX_ROOT)")
endif()

if (GRL_BUILD_GRLX)
  message("** Building grlx addon")
  include_directories(${GRLX_INCLUDE_DIR})
  add_library(${TARGET} SHARED ${SRC}/grlx.cpp)
  target_link_libraries(${TARGET} ${GRLX_LIBRARY})
  grl_link_libraries(${TARGET} base)
  install(TARGETS ${TARGET} DESTINATION ${GRL_LIB_DESTINATION})
  install(DIRECTORY ${SRC}/../include/grl DESTINATION ${GRL_INCLUDE_DESTINATION} FILES_MATCHING PATTERN "*.h")
endif()
