// oracle/ref_prelude.h -- force-included when compiling the REFERENCE's own sources.
// `while(getline(fin, line)>0)` (inputReader/readLoader.cpp:86, matePair/matePair.cpp:83)
// compiled with pre-C++11 libstdc++ through istream -> void*; modern g++ has no such
// conversion.  This operator restores the old meaning ("stream still good").
#pragma once
#include <istream>
inline bool operator>(std::istream& is, int) { return !is.fail(); }
