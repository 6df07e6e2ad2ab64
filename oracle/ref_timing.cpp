// ORACLE-SIDE TEST INFRASTRUCTURE.  Stand-in for src/util/TimeMeasurement.cpp (the reference's wall-clock profiler, out of scope —
// SURVEY §2 row 18): the class declared in the reference's own util/TimeMeasurement.h, implemented so that every scope announces its
// label through `ref_scope_hook` when it begins (+1) and ends (-1).  The reference brackets its pipeline stages with these scopes
// ("FullSystem::trackNewCoarseNoIMU", "FullSystemOptimize", "makeKeyframeChangeTrackingRef", ...), which lets oracle/ref_glue.cpp record
// the inputs and outputs of trackNewCoarse / optimize / setCoarseTrackingRef of an unmodified FullSystem run.  No timing is kept.
#include "util/TimeMeasurement.h"

extern "C" void (*ref_scope_hook)(const char* name, int phase) = nullptr;

namespace dmvio
{
bool TimeMeasurement::saveFileOpen = false;
std::ofstream TimeMeasurement::saveFile;
std::map<std::string, MeasurementLog> TimeMeasurement::logs;

void MeasurementLog::addMeasurement(double) {}
void MeasurementLog::writeLogLine(std::ostream&) const {}
int MeasurementLog::getNum() const { return num; }
double MeasurementLog::getMax() const { return max; }
double MeasurementLog::getMean() const { return 0; }
double MeasurementLog::getVariance() const { return 0; }

TimeMeasurement::TimeMeasurement(std::string name_) : name(name_)
{
	if (ref_scope_hook) ref_scope_hook(name.c_str(), +1);
}
TimeMeasurement::~TimeMeasurement() { end(); }
double TimeMeasurement::end()
{
	if (ended) return 0;
	ended = true;
	if (ref_scope_hook) ref_scope_hook(name.c_str(), -1);
	return 0;
}
void TimeMeasurement::cancel() { ended = true; }
void TimeMeasurement::saveResults(std::string) {}
}
std::ostream& operator<<(std::ostream& os, const dmvio::MeasurementLog&) { return os; }
