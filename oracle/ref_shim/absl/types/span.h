// TEST INFRASTRUCTURE (oracle/_ref build only): hash_filter.h:23 includes this header and uses
// nothing of it.
#pragma once
