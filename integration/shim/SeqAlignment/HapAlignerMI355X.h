// Include-path shim used only by oracle/Makefile's `dropin` target: lets test code include the adapter as
// "SeqAlignment/HapAlignerMI355X.h", i.e. as if integration/HapAlignerMI355X.h had been copied into the HipSTR tree.
#include "../../HapAlignerMI355X.h"
