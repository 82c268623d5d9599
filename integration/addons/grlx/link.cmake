target_link_libraries(${TARGET} addon_grlx)
grl_link_libraries(${TARGET} base)
