#include "PinholeCamera.h"
