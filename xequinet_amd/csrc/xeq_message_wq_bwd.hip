// Reverse half of xeq_message_wq.hip: the same source with XEQ_WQ_PART_BWD (see the note at its top), built with the default
// machine scheduler.
#define XEQ_WQ_PART_BWD 1
#include "xeq_message_wq.hip"
