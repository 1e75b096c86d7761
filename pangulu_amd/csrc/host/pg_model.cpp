// pg_model.cpp -- structure-only roofline model of a rank's task list.
#include "pg_host.h"

namespace pg
{

void compute_task_model(Solver &S)
{
    (void)S;
}

} // namespace pg
