// TEST INFRASTRUCTURE (oracle/_ref build only): cuckoohash_map.hpp:43 only needs this
// include to resolve; it uses no glog symbol.
#pragma once
#include <iostream>
