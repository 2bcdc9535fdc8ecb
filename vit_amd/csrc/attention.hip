// placeholder, replaced below
#include "common.h"
