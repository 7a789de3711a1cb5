/* placeholder: multi-threaded CPU-baseline driver is added below the oracle */
#include "sina_oracle.h"
