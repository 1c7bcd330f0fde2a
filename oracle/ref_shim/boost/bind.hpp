// oracle/ref_shim/boost/bind.hpp -- TEST INFRASTRUCTURE.  src/util/settings.cpp includes <boost/bind.hpp> and uses nothing of it; Boost is not
// installed here.  With this empty header the file compiles unmodified into oracle/_ref/libref.so (see oracle/Makefile, ref_glue.cpp).
#pragma once
