// sched_shim.cpp — std::thread stand-in for the reference's fiber scheduler, ONLY for building the
// reference's runtime objects into oracle/_ref (test infrastructure).  Implements the interface of
// include/lingodb/scheduler/Scheduler.h:29-42 (src/scheduler/Scheduler.cpp needs Boost.Context and
// MLIR headers, neither available offline).  Semantics kept: N workers run one Task to
// completion (setup → {allocateWork → performWork}* → teardown), awaitChildTask may be called
// from a worker (it runs the child task on the calling thread plus helper threads).
#include <mutex>
#include "lingodb/scheduler/Scheduler.h"
#include "lingodb/runtime/ExecutionContext.h"

#include <atomic>
#include <cstdlib>
#include <thread>
#include <vector>

namespace {
size_t g_workers = 0;
thread_local size_t t_worker = 0;

void runOn(lingodb::scheduler::Task* task, size_t workerId) {
   size_t saved = t_worker;
   t_worker = workerId;
   task->setup();
   while (task->hasWork()) {
      if (task->allocateWork()) task->performWork();
   }
   task->teardown();
   t_worker = saved;
}
thread_local bool t_is_worker = false;
void runTask(lingodb::scheduler::Task* task) {
   // From a non-worker thread (awaitEntryTask): all workers are fresh threads and the caller only
   // waits — Task::teardown() clears the worker's thread-local ExecutionContext, which must not
   // happen to the caller.  From a worker (nested awaitChildTask): the worker takes part.
   const bool nested = t_is_worker;
   const size_t self = t_worker;
   std::vector<std::thread> helpers;
   for (size_t w = 0; w < g_workers; w++) {
      if (nested && w == self) continue;
      helpers.emplace_back([task, w]() {
         t_is_worker = true;
         runOn(task, w);
      });
   }
   if (nested) {
      // the calling worker takes part in the child task; Task::teardown() clears the thread's current
      // ExecutionContext, which the rest of the PARENT task on this thread still needs (the reference's
      // fibers give every task its own stack: parallelSort continues after its child tasks)
      auto* saved = lingodb::runtime::getCurrentExecutionContext();
      runOn(task, self);
      lingodb::runtime::setCurrentExecutionContext(saved);
   }
   for (auto& h : helpers) h.join();
}
} // namespace

namespace lingodb::scheduler {
SystemContext::~SystemContext() {}
SchedulerHandle::SchedulerHandle() {}
SchedulerHandle::~SchedulerHandle() {}
std::unique_ptr<SchedulerHandle> startScheduler(size_t numWorkers) {
   if (numWorkers == 0) {
      const char* env = std::getenv("LINGODB_PARALLELISM"); // same knob as Scheduler.cpp:933
      numWorkers = env ? (size_t) std::atoll(env) : std::thread::hardware_concurrency();
   }
   if (numWorkers == 0) numWorkers = 1;
   g_workers = numWorkers;
   return std::make_unique<SchedulerHandle>();
}
void awaitEntryTask(std::unique_ptr<Task> task) { runTask(task.get()); }
void awaitChildTask(std::unique_ptr<Task> task) { runTask(task.get()); }
void enqueueTask(std::unique_ptr<Task> task) { runTask(task.get()); }
size_t getNumWorkers() { return g_workers ? g_workers : 1; }
size_t currentWorkerId() { return t_worker; }
SystemContext& getSystemContext() {
   static SystemContext ctx;
   return ctx;
}
} // namespace lingodb::scheduler
