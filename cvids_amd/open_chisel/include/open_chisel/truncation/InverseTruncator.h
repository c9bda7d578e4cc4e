#include "Truncator.h"
