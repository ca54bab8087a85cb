// TEST INFRASTRUCTURE: lockstep wavefront emulator runtime (64 cooperative fibers, x86-64).
#include <cstdio>
#include <cstdlib>
#include <execinfo.h>
#include <functional>
#include <signal.h>
#include <unistd.h>

#include <jaco/wave_ops.h>

int emu_cur_lane = 0;
int emu_block = 0;
int emu_grid = 1;
EmuWord emu_x[2][64];
unsigned emu_cnt[64];

extern "C" void emu_switch(void** save_sp, void* load_sp);
asm(".text\n.globl emu_switch\n.type emu_switch,@function\nemu_switch:\n"
    " pushq %rbp\n pushq %rbx\n pushq %r12\n pushq %r13\n pushq %r14\n pushq %r15\n"
    " movq %rsp, (%rdi)\n movq %rsi, %rsp\n"
    " popq %r15\n popq %r14\n popq %r13\n popq %r12\n popq %rbx\n popq %rbp\n ret\n");

static const size_t STACK = 1 << 20;
static void* g_sp[64];
static void* g_main_sp;
static char* g_stack[64];
static std::function<void()>* g_body;
static int g_done;

static void* g_site[64][2];
__attribute__((noinline)) void emu_collective() {
  int prev = emu_cur_lane, next = (prev + 1) & 63;
  g_site[prev][0] = __builtin_return_address(0);
  g_site[prev][1] = __builtin_frame_address(0) ? __builtin_return_address(1) : nullptr;
  emu_cur_lane = next;
  emu_switch(&g_sp[prev], g_sp[next]);
}
static void trampoline() {
  (*g_body)();
  int me = emu_cur_lane;
  g_done++;
  if (g_done == 64) {
    emu_switch(&g_sp[me], g_main_sp);
  } else {
    emu_cur_lane = (me + 1) & 63;
    emu_switch(&g_sp[me], g_sp[emu_cur_lane]);
  }
  fprintf(stderr, "wave_emu: finished lane %d resumed (lanes diverged at a cross-lane op)\n", me);
  for (int l = 0; l < 64; l++) fprintf(stderr, "  lane %2d last collective at %p <- %p\n", l, g_site[l][0], g_site[l][1]);
  abort();
}
static void segv_handler(int) {
  void* bt[32];
  int n = backtrace(bt, 32);
  fprintf(stderr, "wave_emu: SIGSEGV in lane %d, block %d\n", emu_cur_lane, emu_block);
  backtrace_symbols_fd(bt, n, 2);
  _exit(139);
}
void emu_run_wave(int block, std::function<void()> body) {
  static bool installed = false;
  if (!installed) {
    static char altstack[1 << 16];
    stack_t ss; ss.ss_sp = altstack; ss.ss_size = sizeof altstack; ss.ss_flags = 0;
    sigaltstack(&ss, nullptr);
    struct sigaction sa; sa.sa_handler = segv_handler; sigemptyset(&sa.sa_mask); sa.sa_flags = SA_ONSTACK;
    sigaction(SIGSEGV, &sa, nullptr);
    installed = true;
  }
  g_body = &body;
  g_done = 0;
  emu_block = block;
  for (int l = 0; l < 64; l++) {
    if (!g_stack[l]) g_stack[l] = (char*)aligned_alloc(64, STACK);
    emu_cnt[l] = 0;
    void** sp = (void**)(g_stack[l] + STACK - 64);
    *--sp = nullptr;               // fake return address of trampoline (keeps 16-byte alignment at entry)
    *--sp = (void*)trampoline;     // ret target
    for (int k = 0; k < 6; k++) *--sp = nullptr;  // rbp rbx r12-r15
    g_sp[l] = sp;
  }
  emu_cur_lane = 0;
  emu_switch(&g_main_sp, g_sp[0]);
}
