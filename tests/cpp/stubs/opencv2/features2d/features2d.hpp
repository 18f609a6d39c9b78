// Declarations only; see tests/cpp/stubs/README.md.
#pragma once
#include <opencv2/core/core.hpp>
