#include "internal.h"
