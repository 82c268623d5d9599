/* main.c -- command line front end of the oracle (TEST INFRASTRUCTURE).
 *
 *   orc_grld [-s seed] [-m libm|portable] [-a sarsa|q] [-t trials] [-b]
 *
 * Prints the test-trial rows of the pendulum tile-coding experiment in the
 * layout of the reference's golden file (tests/template/pendulum-sarsa-tc-0.txt,
 * written by `grld -s 1 tests/pendulum-sarsa-tc.yaml`, bin/runtests.py:21).
 * With -b it prints a timing line instead (used as the CPU baseline).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "oracle.h"

static double now(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

int main(int argc, char **argv)
{
  long seed = 1;
  int trials = 2000, bench = 0;
  orc_spec spec;
  orc_spec_pendulum_sarsa(&spec);

  for (int i = 1; i < argc; ++i)
  {
    if (!strcmp(argv[i], "-s") && i + 1 < argc) seed = atol(argv[++i]);
    else if (!strcmp(argv[i], "-t") && i + 1 < argc) trials = atoi(argv[++i]);
    else if (!strcmp(argv[i], "-m") && i + 1 < argc) spec.math = !strcmp(argv[++i], "portable") ? ORC_MATH_PORTABLE : ORC_MATH_LIBM;
    else if (!strcmp(argv[i], "-a") && i + 1 < argc) spec.agent = !strcmp(argv[++i], "q") ? ORC_AGENT_Q : ORC_AGENT_SARSA;
    else if (!strcmp(argv[i], "-b")) bench = 1;
    else { fprintf(stderr, "usage: %s [-s seed] [-m libm|portable] [-a sarsa|q] [-t trials] [-b]\n", argv[0]); return 1; }
  }

  double t0 = now();
  orc_exp *e = orc_create(&spec, seed);
  if (!e) { fprintf(stderr, "orc_create failed\n"); return 1; }
  double t1 = now();

  int max_rows = trials + 1;
  orc_row *rows = (orc_row *)malloc(sizeof(orc_row) * (size_t)max_rows);
  int n = orc_run(e, trials, rows, max_rows, NULL, 0, NULL);
  double t2 = now();

  if (bench)
  {
    orc_stats st;
    orc_get_stats(e, &st);
    double steps = (double)(st.learn_steps + st.test_steps);
    printf("{\"env_steps\": %.0f, \"learn_steps\": %llu, \"test_steps\": %llu, \"init_s\": %.6f, \"run_s\": %.6f, "
           "\"steps_per_s\": %.1f, \"weight_reads_per_learn_step\": %.3f, \"weight_rmws_per_learn_step\": %.3f, "
           "\"mean_trace_len\": %.3f}\n",
           steps, (unsigned long long)st.learn_steps, (unsigned long long)st.test_steps, t1 - t0, t2 - t1,
           steps / (t2 - t1),
           (double)st.weight_reads / (double)st.learn_steps, (double)st.weight_rmws / (double)st.learn_steps,
           (double)st.trace_entries_sum / (double)st.learn_steps);
  }
  else
  {
    char buf[128];
    for (int i = 0; i < n; ++i)
    {
      orc_format_row(&rows[i], buf, sizeof(buf));
      fputs(buf, stdout);
    }
  }
  free(rows);
  orc_destroy(e);
  return 0;
}
