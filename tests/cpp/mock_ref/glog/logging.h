// tests/cpp/mock_ref — TEST INFRASTRUCTURE: the minimum of glog / OpenCV / ROS messages / the reference's own class declarations that
// emba_amd/host/legm_adapter.hpp touches, so that a compiler (not a reader) checks the adapter against the signatures of
// reference include/emba/model.h:76-108 (tests/test_adapter_syntax.py, g++ -fsyntax-only).  Nothing here is built into anything.
#pragma once
#include <cstdlib>
#include <iostream>
struct MockLogFatal { ~MockLogFatal() { std::abort(); } template <class T> MockLogFatal& operator<<(const T& v) { std::cerr << v; return *this; } };
#define FATAL 3
#define LOG(severity) MockLogFatal()
#define CHECK(cond) if (!(cond)) MockLogFatal() << "CHECK failed: " #cond " "
